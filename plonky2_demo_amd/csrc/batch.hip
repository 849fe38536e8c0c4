// Device-resident PolynomialBatch: the reference's FRI oracle (plonky2/src/fri/oracle.rs:30-133).
//
// HBM layout (all u64, canonical):
//   coeffs  [ncols][n]        coefficient form            (oracle.rs:32 `polynomials`)
//   lde     [ncols][N]        values on 7*H_N, NATURAL order (index i <-> 7*w_N^i), N = n << rate_bits
//   digests level-ordered Merkle digests; leaf j of the tree is LDE row bitrev(j)  (oracle.rs:83-84)
// The reference materialises row-major, bit-reversed `merkle_tree.leaves` (N x ncols); here rows are
// gathered on demand (get_leaf / get_lde_values) and later kernels read the natural-order columns.
#include "context.hpp"
#include <cstring>


// `values`: column-major VALUES to interpolate into b->coeffs (may be b->coeffs itself), or null when b->coeffs already
// holds coefficients
static int batch_commit(gl_ctx* c, gl_batch* b, const gl_t* values) {
    // values -> coefficients (oracle.rs:51-55)
    if (values) {
        c->timing_begin("IFFT");
        int st = gl_ntt_run(c, values, b->n, (uint32_t)b->n, b->coeffs, b->n, b->degree_log, (uint32_t)b->ncols,
                            true, 0, 0, gl_host_inverse_2exp(b->degree_log));
        c->timing_end();
        GL_TRY(st);
    }
    // zero-pad x 2^rate_bits and evaluate on the coset 7*H (oracle.rs:100-125)
    c->timing_begin("FFT + blinding");
    int st_lde = gl_ntt_run(c, b->coeffs, b->n, (uint32_t)b->n, b->lde, b->N(), b->degree_log + b->rate_bits, (uint32_t)b->ncols,
                            false, GL_MULT_GENERATOR, 0, 1);
    c->timing_end();
    GL_TRY(st_lde);
    std::vector<uint64_t> offs(b->ncols);
    for (size_t e = 0; e < b->ncols; e++) offs[e] = e * b->N();
    GL_TRY(gl_merkle_build(c, b->lde, offs.data(), (uint32_t)b->ncols, b->degree_log + b->rate_bits, b->cap_height, &b->tree));
    return GL_OK;
}

extern "C" void gl_batch_free(gl_batch* b);
static int batch_alloc(gl_ctx* c, size_t ncols, size_t n, uint32_t rate_bits, uint32_t cap_height, gl_batch** out) {
    GL_REQUIRE(c && out && ncols >= 1 && n >= 1, GL_ERR_ARG, "PolynomialBatch: bad argument");
    uint32_t lg = 0;
    while ((size_t(1) << lg) < n) lg++;
    GL_REQUIRE((size_t(1) << lg) == n, GL_ERR_ARG, "polynomial length must be a power of two");
    GL_REQUIRE(lg + rate_bits <= 24, GL_ERR_ARG, "LDE size unsupported");
    GL_REQUIRE(cap_height <= lg + rate_bits, GL_ERR_ARG, "cap_height should be at most log2(leaves.len())");
    GL_TRY(c->activate());
    gl_batch* b = new gl_batch();
    b->ctx = c; c->retain(); b->ncols = ncols; b->n = n; b->degree_log = lg; b->rate_bits = rate_bits; b->cap_height = cap_height;
    int st = c->pool_alloc(ncols * n * sizeof(gl_t), (void**)&b->coeffs);
    if (st == GL_OK) st = c->pool_alloc(ncols * b->N() * sizeof(gl_t), (void**)&b->lde);
    if (st != GL_OK) { gl_batch_free(b); return st; }      // nothing leaks when the device is out of memory
    *out = b;
    return GL_OK;
}

static int batch_from_host(gl_ctx* c, const uint64_t* const* h_cols, size_t ncols, size_t n, uint32_t rate_bits, uint32_t blinding,
                           uint32_t cap_height, bool is_values, gl_batch** out) {
    GL_REQUIRE(h_cols, GL_ERR_ARG, "PolynomialBatch: null columns");
    GL_REQUIRE(blinding == 0, GL_ERR_UNSUPPORTED, "blinding (zero-knowledge salts) is not supported");
    gl_batch* b = nullptr;
    GL_TRY(batch_alloc(c, ncols, n, rate_bits, cap_height, &b));
    for (size_t col = 0; col < ncols; col++) {
        if (!h_cols[col]) { gl_batch_free(b); return gl_fail(GL_ERR_ARG, "null column", __FILE__, __LINE__); }
        hipError_t e = hipMemcpyAsync(b->coeffs + col * n, h_cols[col], n * sizeof(gl_t), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { gl_batch_free(b); return gl_fail(GL_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); }
    }
    GL_CHECK_HIP(gl_stream_wait(c->stream));   // caller-owned pageable columns
    int st = batch_commit(c, b, is_values ? b->coeffs : nullptr);
    if (st != GL_OK) { gl_batch_free(b); return st; }
    *out = b;
    return GL_OK;
}

extern "C" int gl_batch_from_values(gl_ctx* c, const uint64_t* const* h_cols, size_t ncols, size_t n, uint32_t rate_bits,
                                    uint32_t blinding, uint32_t cap_height, gl_batch** out) {
    return batch_from_host(c, h_cols, ncols, n, rate_bits, blinding, cap_height, true, out);
}
extern "C" int gl_batch_from_coeffs(gl_ctx* c, const uint64_t* const* h_cols, size_t ncols, size_t n, uint32_t rate_bits,
                                    uint32_t blinding, uint32_t cap_height, gl_batch** out) {
    return batch_from_host(c, h_cols, ncols, n, rate_bits, blinding, cap_height, false, out);
}
extern "C" int gl_batch_from_device(gl_ctx* c, const uint64_t* d_cols, size_t ncols, size_t n, uint32_t rate_bits,
                                    uint32_t cap_height, int is_values, gl_batch** out) {
    GL_REQUIRE(d_cols, GL_ERR_ARG, "PolynomialBatch: null device columns");
    gl_batch* b = nullptr;
    GL_TRY(batch_alloc(c, ncols, n, rate_bits, cap_height, &b));
    if (!is_values) {      // coefficients are kept: copy; values are interpolated straight out of the caller's matrix (left untouched)
        hipError_t e = hipMemcpyAsync(b->coeffs, d_cols, ncols * n * sizeof(gl_t), hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) { gl_batch_free(b); return gl_fail(GL_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); }
    }
    int st = batch_commit(c, b, is_values ? (const gl_t*)d_cols : nullptr);
    if (st != GL_OK) { gl_batch_free(b); return st; }
    *out = b;
    return GL_OK;
}

extern "C" int gl_batch_cap(const gl_batch* b, uint64_t* h_out) {
    GL_REQUIRE(b && h_out, GL_ERR_ARG, "null argument");
    const GlMerkle& m = b->tree;
    return gl_copy_d2h(b->ctx, h_out, m.level_ptr(m.num_levels() - 1), (size_t(4) << m.cap_height) * sizeof(gl_t));
}

// out[c] = lde[c][row]
__global__ void k_gather_row(const gl_t* lde, uint64_t stride, uint32_t ncols, uint64_t row, gl_t* out) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncols) out[c] = lde[c * stride + row];
}
static size_t host_bitrev(size_t x, uint32_t bits) {
    size_t r = 0;
    for (uint32_t i = 0; i < bits; i++) r = (r << 1) | ((x >> i) & 1);
    return r;
}
static int batch_row(const gl_batch* b, size_t natural_row, uint64_t* h_out) {
    gl_ctx* c = b->ctx;
    GL_TRY(c->activate());
    GL_TRY(c->ensure_dev_small(b->ncols * sizeof(gl_t)));
    hipLaunchKernelGGL(k_gather_row, dim3((unsigned)((b->ncols + 127) / 128)), dim3(128), 0, c->stream, b->lde, (uint64_t)b->N(),
                       (uint32_t)b->ncols, (uint64_t)natural_row, c->dev_small);
    GL_CHECK_HIP(hipGetLastError());
    return gl_copy_d2h(c, h_out, c->dev_small, b->ncols * sizeof(gl_t));
}
extern "C" int gl_batch_get_leaf(const gl_batch* b, size_t leaf_index, uint64_t* h_out) {
    GL_REQUIRE(b && h_out && leaf_index < b->N(), GL_ERR_ARG, "bad leaf index");
    return batch_row(b, host_bitrev(leaf_index, b->degree_log + b->rate_bits), h_out);
}
extern "C" int gl_batch_get_lde_values(const gl_batch* b, size_t index, size_t step, uint64_t* h_out) {
    GL_REQUIRE(b && h_out && index * step < b->N(), GL_ERR_ARG, "bad LDE index");
    // oracle.rs:128-133 reads leaves[reverse_bits(index*step)]; leaf j is LDE row bitrev(j), so this is row index*step
    return batch_row(b, index * step, h_out);
}
extern "C" int gl_batch_prove(const gl_batch* b, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings) {
    GL_REQUIRE(b && h_out, GL_ERR_ARG, "null argument");
    return gl_merkle_prove_impl(b->ctx, b->tree, leaf_index, h_out, n_siblings);
}
extern "C" int gl_batch_coeffs(const gl_batch* b, uint64_t* h_out) {
    GL_REQUIRE(b && h_out, GL_ERR_ARG, "null argument");
    return gl_copy_d2h(b->ctx, h_out, b->coeffs, b->ncols * b->n * sizeof(gl_t));
}
extern "C" int gl_batch_lde(const gl_batch* b, uint64_t* h_out) {
    GL_REQUIRE(b && h_out, GL_ERR_ARG, "null argument");
    return gl_copy_d2h(b->ctx, h_out, b->lde, b->ncols * b->N() * sizeof(gl_t));
}
extern "C" size_t gl_batch_ncols(const gl_batch* b) { return b ? b->ncols : 0; }
extern "C" size_t gl_batch_degree(const gl_batch* b) { return b ? b->n : 0; }
extern "C" const uint64_t* gl_batch_dev_coeffs(const gl_batch* b) { return b ? b->coeffs : nullptr; }
extern "C" const uint64_t* gl_batch_dev_lde(const gl_batch* b) { return b ? b->lde : nullptr; }
extern "C" void gl_batch_free(gl_batch* b) {
    if (!b) return;
    gl_merkle_release(b->ctx, &b->tree);
    if (b->coeffs) b->ctx->pool_release(b->coeffs);
    if (b->lde) b->ctx->pool_release(b->lde);
    gl_ctx_release(b->ctx);
    delete b;
}
