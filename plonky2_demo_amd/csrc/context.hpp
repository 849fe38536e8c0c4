// Library context: one per (device, stream).  Owns twiddle / power tables and scratch memory so that
// no entry point allocates or synchronises on its hot path (graph-capture friendly after warm-up).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <chrono>
#include <time.h>
#include <sys/prctl.h>
#include <map>
#include <tuple>
#include <mutex>
#include <unordered_map>
#include <string>
#include <utility>
#include <vector>
#include "gl64.cuh"
#include "../../include/plonky2_mi355x.h"

struct gl_ctx;
void gl_ctx_release(gl_ctx* c);      // drops one reference (ntt.hip)
extern thread_local std::string g_gl_last_error;
int gl_fail(int code, const char* what, const char* file, int line);

#define GL_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return gl_fail(GL_ERR_HIP, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define GL_REQUIRE(cond, code, msg)                                      \
    do {                                                                 \
        if (!(cond)) return gl_fail(code, msg, __FILE__, __LINE__);      \
    } while (0)
#define GL_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != GL_OK) return _s;  \
    } while (0)

// Wait for a stream by polling it.  hipStreamSynchronize follows the device's scheduling flags, and a process that has brought
// up RCCL blocks on an interrupt there: every one of the ~10 transcript round trips of a proof then pays a wake-up latency
// (measured: 245 -> 210 proofs/s with a one-rank process group).  Polling is independent of what other libraries set.
// One or two waiting threads spin (a lone proof's latency is the sum of ~15 such waits, and a sleeping poll adds ~35 us to
// each: 6.3 -> 6.8 ms measured).  With more waiters -- many proofs in flight -- a wait lasts milliseconds, and spinning threads
// exhaust the CPU quota (measured on the 1-GPU box, 16 lanes: the cgroup throttled the process in half of its scheduling
// periods, and 24 or 32 lanes ran 20-40 % SLOWER than 16) and keep threads that have work to submit off the cores when several
// ranks share a host: after ~50 us those threads sleep ~20 us between polls.
inline std::atomic<int> gl_stream_waiters{0};
// proofs currently inside gl_prove* in this process (any context): with several in flight the GPU's issue slots are the bound
// and latency-hiding variants that spend more instructions per result stop paying (merkle.hip gl_coop_max_nodes)
inline std::atomic<int> gl_proofs_in_flight{0};
inline hipError_t gl_stream_wait(hipStream_t s) {
    hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    static thread_local bool slack_set = false;
    struct Count { Count() { gl_stream_waiters.fetch_add(1, std::memory_order_relaxed); } ~Count() { gl_stream_waiters.fetch_sub(1, std::memory_order_relaxed); } } count;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (gl_stream_waiters.load(std::memory_order_relaxed) <= 2 || std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(50)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            // (the first sleeping wait of a thread lowers ITS timer slack from the default 50 us to 1 us; nothing else is changed)
            if (!slack_set) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); slack_set = true; }
            struct timespec ts = {0, 20000};
            (void)nanosleep(&ts, nullptr);
        }
    }
}

static const gl_t GL_MULT_GENERATOR = 7;                          // field/src/goldilocks_field.rs:80
static const gl_t GL_POW2_GENERATOR = 1753635133440165772ULL;     // field/src/goldilocks_field.rs:87

inline gl_t gl_host_root_of_unity(unsigned lg) {                   // field/src/types.rs:268-272
    gl_t r = GL_POW2_GENERATOR;
    for (unsigned i = lg; i < 32; i++) r = gl_sqr(r);
    return gl_canon(r);
}
inline gl_t gl_host_inverse_2exp(unsigned e) { return GL_P - ((GL_P - 1) >> e); }   // types.rs:227-266

struct GlPowTable {
    gl_t* lo = nullptr;   // base^j, j < 2048
    gl_t* hi = nullptr;   // scale * base^(2048 j)
    uint32_t hi_len = 0;
};

struct gl_ctx {
    // Lifetime: one reference for the creator (dropped by gl_ctx_destroy) plus one per live handle that points at this context
    // (gl_batch, gl_merkle, gl_circuit, gl_fri, gl_matmul_witgen).  The context -- its stream, tables and allocator -- is torn
    // down when the last reference goes, so handles may be freed after gl_ctx_destroy in any order.
    std::atomic<int> refs{1};
    void retain() { refs.fetch_add(1, std::memory_order_relaxed); }
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    gl_t* tw_local[2] = {nullptr, nullptr};                       // w_4096^e forward / inverse
    std::map<std::pair<gl_t, gl_t>, GlPowTable> pow_tables;       // (base, scale) -> table
    std::map<std::tuple<gl_t, gl_t, uint32_t>, gl_t*> pass_tables; // (w_N, scale, lgN1) -> scale * inter-pass twiddles in column-pass output order
    int get_pass_table(gl_t w, gl_t scale, uint32_t lgN1, uint32_t lgN2, const gl_t** out);
    gl_t* scratch = nullptr;
    size_t scratch_elems = 0;
    // inter-pass scratch: at most 2^26 elements (512 MiB), i.e. 64 polynomials of 2^20 per column/row launch pair.  Fewer, larger
    // launches beat keeping the scratch Infinity-Cache resident: forward 2^20 x 64 takes 0.829 / 0.751 / 0.737 / 0.722 ms with
    // 2^23 / 2^24 / 2^25 / 2^26 elements (8 / 4 / 2 / 1 launch pairs); prove() is indifferent (+0.6 %)
    size_t scratch_target = size_t(1) << 26;
    // pinned staging buffers for small device-to-host results (caps, openings, query rows).  A context's stream may be used by
    // several host threads at once (proofs on other contexts read the shared circuit's batches through ITS context), so a
    // buffer is taken from this list for one copy and handed back afterwards
    std::mutex pin_mu;
    std::vector<std::pair<void*, size_t>> pin_free;
    int pin_acquire(size_t bytes, void** out, size_t* cap);
    void pin_release(void* p, size_t cap);
    gl_t* dev_small = nullptr;                                    // 1 MiB device staging
    size_t dev_small_bytes = 0;

    // optional per-launch timing (HIP events on the ctx stream); scope names follow the reference's
    // TimingTree labels where one exists (plonky2/src/util/timing.rs, fri/oracle.rs:51-89)
    struct TimingRec { const char* name; hipEvent_t start, stop; };
    bool timing_enabled = false;
    std::vector<TimingRec> timing_recs;
    std::vector<size_t> timing_stack;       // open scopes (scopes nest like the reference's TimingTree)
    void timing_begin(const char* name);
    void timing_end();

    // Stream-ordered caching allocator: prove() allocates the same buffer sizes for every proof, and hipMalloc /
    // hipFree synchronise the whole device (which would serialise proofs running on other streams).  Blocks are
    // recycled within this context only; all work of a context is ordered on its one stream, so a block may be
    // handed out again without waiting for the kernels that last used it.
    std::mutex pool_mu;
    std::multimap<size_t, void*> pool_free_blocks;
    std::unordered_map<void*, size_t> pool_block_size;
    size_t pool_bytes = 0;
    int pool_alloc(size_t bytes, void** out);
    void pool_release(void* p);
    void pool_trim();

    // parity tests: keep the Z / partial-product values and the quotient chunks of each proof on the host
    bool capture_intermediates = false;

    int activate();
    int ensure_scratch(size_t elems);
    int ensure_dev_small(size_t bytes);
    int get_pow_table(gl_t base, gl_t scale, uint32_t hi_len, GlPowTable* out);
    std::vector<gl_t*> retired_tables;      // superseded (shorter) power tables: possibly still in flight, freed with the context
    // small read-only index tables (Merkle leaf element offsets, ...) cached by content
    std::map<std::vector<uint64_t>, uint64_t*> offset_tables;
    int get_offsets_table(const uint64_t* host, size_t len, const uint64_t** d_out);
};

// RAII scope of the optional per-launch timing
struct GlTimed {
    gl_ctx* c;
    GlTimed(gl_ctx* ctx, const char* name) : c(ctx) { c->timing_begin(name); }
    ~GlTimed() { c->timing_end(); }
};

// ---- NTT launcher (ntt.hip) ---------------------------------------------------------------------------
// dst[b][k] = post_const * post_shift^k * sum_i (pre_shift^i * src[b][i]) * w^(+-ik),  i < n_in, k < 2^lgN
// pre_shift / post_shift == 0 mean "no scaling".  src may equal dst.  Output canonical.
int gl_ntt_run(gl_ctx* ctx, const gl_t* src, uint64_t src_stride, uint32_t n_in, gl_t* dst, uint64_t dst_stride,
               uint32_t lgN, uint32_t batch, bool inverse, gl_t pre_shift, gl_t post_shift, gl_t post_const);

// ---- Merkle (merkle.hip) ------------------------------------------------------------------------------
struct GlMerkle {
    uint32_t lg_leaves = 0, cap_height = 0, leaf_len = 0;
    gl_t* digests = nullptr;                 // levels concatenated, level l at level_off[l] (in digests)
    std::vector<uint64_t> level_off;         // in units of digests (4 x u64)
    uint64_t total_digests = 0;
    gl_t* level_ptr(uint32_t l) const { return digests + 4 * level_off[l]; }
    uint32_t num_levels() const { return lg_leaves - cap_height + 1; }
};
// leaf r (natural order) has elements base[offsets[e] + r], e < leaf_len; it is leaf bitrev(r) of the tree
int gl_merkle_build(gl_ctx* ctx, const gl_t* base, const uint64_t* host_offsets, uint32_t leaf_len,
                    uint32_t lg_leaves, uint32_t cap_height, GlMerkle* out);
void gl_merkle_release(gl_ctx* ctx, GlMerkle* m);
int gl_merkle_prove_impl(gl_ctx* c, const GlMerkle& m, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings);

// ---- PolynomialBatch (batch.hip) ------------------------------------------------------------------------
struct gl_batch {
    gl_ctx* ctx = nullptr;
    size_t ncols = 0, n = 0;
    uint32_t degree_log = 0, rate_bits = 0, cap_height = 0;
    gl_t* coeffs = nullptr;      // [ncols][n]
    gl_t* lde = nullptr;         // [ncols][N], natural order (index i <-> 7 * w_N^i)
    GlMerkle tree;
    size_t N() const { return n << rate_bits; }
};

// RAII block from a context's stream-ordered pool
struct DevBuf {
    gl_ctx* c; void* p = nullptr;
    explicit DevBuf(gl_ctx* ctx) : c(ctx) {}
    int alloc(size_t bytes) { return c->pool_alloc(bytes, &p); }
    void release() { if (p) { c->pool_release(p); p = nullptr; } }
    ~DevBuf() { release(); }
    template <class T> T* as() const { return (T*)p; }
};
