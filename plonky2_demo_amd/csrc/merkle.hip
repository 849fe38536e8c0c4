// Poseidon Merkle trees on gfx950.
// Replaces MerkleTree::new / fill_subtree (plonky2/src/hash/merkle_tree.rs:69-165), MerkleTree::prove
// (:171-207) and the leaf `transpose` + `reverse_index_bits_in_place` of PolynomialBatch::from_coeffs
// (plonky2/src/fri/oracle.rs:83-84): leaves are hashed straight out of the column-major, natural-order
// LDE buffers (lane r reads element e of row r at base[offsets[e] + r] -> fully coalesced 8-byte
// streams) and the bit reversal is applied only to where the 32-byte digest is written.  The reference's
// interleaved digest array (merkle_tree.rs:43-51) is not observable; digests are stored level by level.
// Hashing is integer-ALU bound (one permutation ~ 1e3 modular multiplies per 64 B of input).
#include "context.hpp"
#include "poseidon.cuh"
#include <cstring>
#include <memory>

struct gl_merkle {
    gl_ctx* ctx = nullptr;
    GlMerkle tree;
    gl_t* leaves = nullptr;      // device copy, row-major [num_leaves][leaf_len] in Merkle order
    size_t num_leaves = 0, leaf_len = 0;
};

#define GL_COOP_MAX_NODES_DEFAULT 8192u      // 8192 hashes x 16 lanes = 2048 waves = 2 per SIMD
#define GL_COOP_MAX_NODES_THROUGHPUT 1024u   // with more than two proofs in flight
// tuning knob (environment GL_COOP_MAX_NODES): launches of at most this many hashes use the 16-lane cooperative permutation
static uint32_t gl_coop_max_nodes() {
    static const long env = [] { const char* e = getenv("GL_COOP_MAX_NODES"); return e ? (long)strtoul(e, nullptr, 10) : -1L; }();
    if (env >= 0) return (uint32_t)env;
    // The 16-lane hash costs 3.5 x the instructions of the lane-per-hash one and buys latency.  With several proofs in flight
    // the other proofs hide that latency and the issue slots are what is scarce: measured with 16 in flight, thresholds of
    // 2048 / 512 / 128 give 287 proofs/s against 281-283 for 8192.  (The tree tops, <= 64 nodes per cap subtree, stay fused.)
    return gl_proofs_in_flight.load(std::memory_order_relaxed) > 2 ? GL_COOP_MAX_NODES_THROUGHPUT : GL_COOP_MAX_NODES_DEFAULT;
}

__device__ __forceinline__ uint32_t d_bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// digest of natural row r -> digests[bitrev(r)].  WPE: waves per SIMD the register allocation must leave room for
template <int WPE>
__global__ __launch_bounds__(256, WPE) void k_merkle_leaves(const gl_t* __restrict__ base, const uint64_t* __restrict__ offsets,
                                                       uint32_t leaf_len, uint32_t lg_leaves, gl_t* __restrict__ digests) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (1u << lg_leaves)) return;
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    if (leaf_len <= 4) {   // hash_or_noop short path: canonical copy (plonk/config.rs:55-62)
        for (uint32_t e = 0; e < leaf_len; e++) s[e] = gl_canon(base[offsets[e] + r]);
    } else {
        for (uint32_t e0 = 0; e0 < leaf_len; e0 += 8) {   // overwrite-mode sponge (hashing.rs:117-131)
            const uint32_t c = leaf_len - e0 < 8 ? leaf_len - e0 : 8;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((uint32_t)i < c) s[i] = base[offsets[e0 + i] + r];
            psd_permute(s);
        }
    }
    const uint32_t j = d_bitrev(r, lg_leaves);
    ulonglong2* out = reinterpret_cast<ulonglong2*>(digests + 4ull * j);
    out[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    out[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// parent[i] = two_to_one(child[2i], child[2i+1])   (hashing.rs:98-115)
__global__ __launch_bounds__(256) void k_merkle_level(const gl_t* __restrict__ child, gl_t* __restrict__ parent, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const ulonglong2* in = reinterpret_cast<const ulonglong2*>(child + 8ull * i);
    ulonglong2 a = in[0], b = in[1], c = in[2], d = in[3];
    gl_t s[12] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, 0, 0, 0, 0};
    psd_permute(s);
    ulonglong2* out = reinterpret_cast<ulonglong2*>(parent + 4ull * i);
    out[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    out[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// the same two kernels with one state per 16 lanes (psd_coop_permute): for launches too small to fill the chip
__global__ __launch_bounds__(256) void k_merkle_level_coop(const gl_t* __restrict__ child, gl_t* __restrict__ parent, uint32_t count) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, node = gid >> 4;
    const int l = (int)(gid & 15);
    if (node >= count) return;                                    // whole 16-lane groups leave together
    gl_t s = l < 8 ? child[8ull * node + l] : 0;
    s = psd_coop_permute(s, l);
    if (l < 4) parent[4ull * node + l] = gl_canon(s);
}
__global__ __launch_bounds__(256) void k_merkle_leaves_coop(const gl_t* __restrict__ base, const uint64_t* __restrict__ offsets,
                                                            uint32_t leaf_len, uint32_t lg_leaves, gl_t* __restrict__ digests) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, r = gid >> 4;
    const int l = (int)(gid & 15);
    if (r >= (1u << lg_leaves)) return;
    gl_t s = 0;
    if (leaf_len <= 4) {   // hash_or_noop short path
        if ((uint32_t)l < leaf_len) s = gl_canon(base[offsets[l] + r]);
    } else {
        for (uint32_t e0 = 0; e0 < leaf_len; e0 += 8) {
            const uint32_t c = leaf_len - e0 < 8 ? leaf_len - e0 : 8;
            if ((uint32_t)l < c) s = base[offsets[e0 + l] + r];
            s = psd_coop_permute(s, l);
        }
    }
    if (l < 4) digests[4ull * d_bitrev(r, lg_leaves) + l] = gl_canon(s);
}

// The top of the tree in ONE launch: the 2^cap_height subtrees under the cap are independent, so workgroup b walks subtree
// b from the level with GL_TOP_NODES nodes per subtree up to its root (a cap entry) with workgroup barriers between levels,
// 16 lanes per hash.  Replaces log2(GL_TOP_NODES) + 1 dependent launches (each ~14 us of work behind a ~8 us dispatch gap).
// digests: level-ordered array, `lev_off[l]` = first digest of level l; first_level = level whose nodes are computed first.
#define GL_TOP_NODES 64u                // 64 hashes x 16 lanes = one 1024-thread workgroup
__global__ __launch_bounds__(1024) void k_merkle_top_coop(gl_t* __restrict__ digests, const uint64_t* __restrict__ lev_off, uint32_t first_level,
                                                          uint32_t num_levels, uint32_t nodes_first /* per subtree at first_level */) {
    const uint32_t sub = blockIdx.x, g = threadIdx.x >> 4;
    const int l = (int)(threadIdx.x & 15);
    uint32_t nodes = nodes_first;
    for (uint32_t lev = first_level; lev < num_levels; lev++, nodes >>= 1) {
        const gl_t* child = digests + 4 * lev_off[lev - 1] + 8ull * sub * nodes;      // 2 * nodes children of this subtree
        gl_t* parent = digests + 4 * lev_off[lev] + 4ull * sub * nodes;
        if (g < nodes) {
            gl_t s = l < 8 ? child[8ull * g + l] : 0;
            s = psd_coop_permute(s, l);
            if (l < 4) parent[4ull * g + l] = gl_canon(s);
        }
        __syncthreads();                                                              // the next level reads what this one wrote
    }
}

__global__ void k_poseidon_states(gl_t* states, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    gl_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    psd_permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[12 * i + k] = gl_canon(s[k]);
}

// hash_or_noop of row-major rows
__global__ void k_hash_rows(const gl_t* rows, size_t count, uint32_t len, gl_t* out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const gl_t* row = rows + r * len;
    if (len <= 4) {
        for (uint32_t e = 0; e < len; e++) s[e] = gl_canon(row[e]);
    } else {
        for (uint32_t e0 = 0; e0 < len; e0 += 8) {
            const uint32_t c = len - e0 < 8 ? len - e0 : 8;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if ((uint32_t)i < c) s[i] = row[e0 + i];
            psd_permute(s);
        }
    }
    for (int i = 0; i < 4; i++) out[4 * r + i] = gl_canon(s[i]);
}

int gl_merkle_build(gl_ctx* c, const gl_t* base, const uint64_t* host_offsets, uint32_t leaf_len, uint32_t lg_leaves,
                    uint32_t cap_height, GlMerkle* out) {
    GL_REQUIRE(c && base && host_offsets && out, GL_ERR_ARG, "gl_merkle_build: null argument");
    GL_REQUIRE(cap_height <= lg_leaves, GL_ERR_ARG, "cap_height should be at most log2(leaves.len())");   // merkle_tree.rs:137-143
    GL_REQUIRE(lg_leaves <= 30 && leaf_len >= 1 && leaf_len <= 4096, GL_ERR_ARG, "gl_merkle_build: unsupported shape");
    GL_TRY(c->activate());
    out->lg_leaves = lg_leaves; out->cap_height = cap_height; out->leaf_len = leaf_len;
    const uint32_t levels = lg_leaves - cap_height + 1;
    out->level_off.resize(levels);
    uint64_t off = 0;
    for (uint32_t l = 0; l < levels; l++) { out->level_off[l] = off; off += uint64_t(1) << (lg_leaves - l); }
    out->total_digests = off;
    GL_TRY(c->pool_alloc(off * 4 * sizeof(gl_t), (void**)&out->digests));
    // offsets -> device: tiny table, cached per content in the context (no sync in steady state)
    const uint64_t* d_off = nullptr;
    GL_TRY(c->get_offsets_table(host_offsets, leaf_len, &d_off));
    const uint32_t n = 1u << lg_leaves;
    const uint32_t coop = gl_coop_max_nodes();                 // one decision per tree
    c->timing_begin("merkle_leaf_hash");
    // below the threshold one-lane-per-hash launches cannot fill the 1024 SIMDs: use 16 lanes per hash (latency / 3)
    if (n <= coop) hipLaunchKernelGGL(k_merkle_leaves_coop, dim3((n * 16 + 255) / 256), dim3(256), 0, c->stream, base, d_off, leaf_len, lg_leaves, out->level_ptr(0));
    else {
        static const int wpe = [] { const char* e = getenv("GL_LEAF_WPE"); return e ? atoi(e) : 5; }();      // tuning knob
        auto kern = wpe >= 8 ? k_merkle_leaves<8> : wpe == 7 ? k_merkle_leaves<7> : wpe == 6 ? k_merkle_leaves<6> : wpe == 4 ? k_merkle_leaves<4> : k_merkle_leaves<5>;
        hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, c->stream, base, d_off, leaf_len, lg_leaves, out->level_ptr(0));
    }
    c->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    c->timing_begin("merkle_levels");
    // levels with at most GL_TOP_NODES nodes per cap subtree are finished by one launch (k_merkle_top_coop)
    uint32_t top_first = levels;
    if (coop > 0)
        for (uint32_t l = 1; l < levels; l++)
            if (((1u << (lg_leaves - l)) >> cap_height) <= GL_TOP_NODES && ((1u << (lg_leaves - l)) >> cap_height) >= 1) { top_first = l; break; }
    const uint64_t* d_lev_off = nullptr;
    if (top_first < levels && levels - top_first >= 2) GL_TRY(c->get_offsets_table(out->level_off.data(), out->level_off.size(), &d_lev_off));
    else top_first = levels;
    for (uint32_t l = 1; l < top_first; l++) {
        const uint32_t cnt = 1u << (lg_leaves - l);
        if (cnt <= coop) hipLaunchKernelGGL(k_merkle_level_coop, dim3((cnt * 16 + 255) / 256), dim3(256), 0, c->stream, out->level_ptr(l - 1), out->level_ptr(l), cnt);
        else hipLaunchKernelGGL(k_merkle_level, dim3((cnt + 255) / 256), dim3(256), 0, c->stream, out->level_ptr(l - 1), out->level_ptr(l), cnt);
    }
    if (top_first < levels)
        hipLaunchKernelGGL(k_merkle_top_coop, dim3(1u << cap_height), dim3(1024), 0, c->stream, out->digests, d_lev_off, top_first, levels,
                           (1u << (lg_leaves - top_first)) >> cap_height);
    c->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
void gl_merkle_release(gl_ctx* c, GlMerkle* m) {
    if (m && m->digests) { c->pool_release(m->digests); m->digests = nullptr; }
}

// --------------------------------------------------------------------------------------------- C ABI
extern "C" int gl_poseidon_permute(gl_ctx* c, uint64_t* d_states, size_t count) {
    GL_REQUIRE(c && d_states, GL_ERR_ARG, "null argument");
    if (!count) return GL_OK;
    GL_TRY(c->activate());
    hipLaunchKernelGGL(k_poseidon_states, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream, d_states, count);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
extern "C" int gl_hash_rows(gl_ctx* c, const uint64_t* d_rows, size_t count, size_t len, uint64_t* d_out) {
    GL_REQUIRE(c && d_rows && d_out && len >= 1, GL_ERR_ARG, "bad argument");
    if (!count) return GL_OK;
    GL_TRY(c->activate());
    hipLaunchKernelGGL(k_hash_rows, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream, d_rows, count, (uint32_t)len, d_out);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}

// row-major [rows][cols] -> column-major [cols][rows] with the row index bit-reversed
__global__ void k_rows_to_natural_cols(const gl_t* rows, uint32_t lg_rows, uint32_t cols, gl_t* out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nrows = uint64_t(1) << lg_rows;
    if (idx >= nrows * cols) return;
    const uint32_t c = (uint32_t)(idx >> lg_rows), r = (uint32_t)(idx & (nrows - 1));
    out[idx] = rows[(uint64_t)d_bitrev(r, lg_rows) * cols + c];
}

extern "C" int gl_merkle_new(gl_ctx* c, const uint64_t* h_leaves, size_t num_leaves, size_t leaf_len, uint32_t cap_height, gl_merkle** out) {
    GL_REQUIRE(c && h_leaves && out && num_leaves >= 1 && leaf_len >= 1, GL_ERR_ARG, "gl_merkle_new: bad argument");
    uint32_t lg = 0;
    while ((size_t(1) << lg) < num_leaves) lg++;
    GL_REQUIRE((size_t(1) << lg) == num_leaves, GL_ERR_ARG, "number of leaves must be a power of two");
    GL_REQUIRE(cap_height <= lg, GL_ERR_ARG, "cap_height should be at most log2(leaves.len())");
    GL_TRY(c->activate());
    std::unique_ptr<gl_merkle, void (*)(gl_merkle*)> t(new gl_merkle(), gl_merkle_free);      // error paths free everything
    t->ctx = c; c->retain(); t->num_leaves = num_leaves; t->leaf_len = leaf_len;
    const size_t bytes = num_leaves * leaf_len * sizeof(gl_t);
    struct Scratch { gl_ctx* c; gl_t* p = nullptr; ~Scratch() { if (p) { (void)gl_stream_wait(c->stream); (void)hipFree(p); } } } cols{c};
    GL_CHECK_HIP(hipMalloc((void**)&t->leaves, bytes));
    GL_CHECK_HIP(hipMalloc((void**)&cols.p, bytes));
    GL_TRY(gl_copy_h2d(c, t->leaves, h_leaves, bytes));
    const uint64_t total = (uint64_t)num_leaves * leaf_len;
    hipLaunchKernelGGL(k_rows_to_natural_cols, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, t->leaves, lg, (uint32_t)leaf_len, cols.p);
    GL_CHECK_HIP(hipGetLastError());
    std::vector<uint64_t> offs(leaf_len);
    for (size_t e = 0; e < leaf_len; e++) offs[e] = e * num_leaves;
    GL_TRY(gl_merkle_build(c, cols.p, offs.data(), (uint32_t)leaf_len, lg, cap_height, &t->tree));
    *out = t.release();
    return GL_OK;
}
extern "C" int gl_merkle_cap(const gl_merkle* t, uint64_t* h_out) {
    GL_REQUIRE(t && h_out, GL_ERR_ARG, "null argument");
    const GlMerkle& m = t->tree;
    return gl_copy_d2h(t->ctx, h_out, m.level_ptr(m.num_levels() - 1), (size_t(4) << m.cap_height) * sizeof(gl_t));
}

__global__ void k_gather_siblings(const gl_t* digests, const uint64_t* level_off, uint32_t levels, uint32_t leaf_index, gl_t* out) {
    const uint32_t l = threadIdx.x >> 2, k = threadIdx.x & 3;
    if (l >= levels) return;
    const uint32_t idx = (leaf_index >> l) ^ 1u;
    out[4 * l + k] = digests[4 * (level_off[l] + idx) + k];
}
int gl_merkle_prove_impl(gl_ctx* c, const GlMerkle& m, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings) {
    GL_REQUIRE(leaf_index < (size_t(1) << m.lg_leaves), GL_ERR_ARG, "leaf index out of range");
    const uint32_t nsib = m.lg_leaves - m.cap_height;
    if (n_siblings) *n_siblings = nsib;
    if (!nsib) return GL_OK;
    GL_TRY(c->activate());
    GL_TRY(c->ensure_dev_small(4096));
    uint64_t* d_off = (uint64_t*)c->dev_small;
    gl_t* d_out = c->dev_small + 64;
    GL_CHECK_HIP(hipMemcpyAsync(d_off, m.level_off.data(), nsib * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_siblings, dim3(1), dim3(4 * nsib), 0, c->stream, m.digests, d_off, nsib, (uint32_t)leaf_index, d_out);
    GL_CHECK_HIP(hipGetLastError());
    return gl_copy_d2h(c, h_out, d_out, nsib * 4 * sizeof(gl_t));
}
extern "C" int gl_merkle_prove(const gl_merkle* t, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings) {
    GL_REQUIRE(t && h_out, GL_ERR_ARG, "null argument");
    return gl_merkle_prove_impl(t->ctx, t->tree, leaf_index, h_out, n_siblings);
}
extern "C" void gl_merkle_free(gl_merkle* t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)gl_stream_wait(t->ctx->stream);
    gl_merkle_release(t->ctx, &t->tree);
    if (t->leaves) (void)hipFree(t->leaves);
    gl_ctx_release(t->ctx);
    delete t;
}
