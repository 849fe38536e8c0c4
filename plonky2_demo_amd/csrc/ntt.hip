// NTT launchers, context management and the NTT / field entry points of the C ABI.
#include "context.hpp"
#include <atomic>
#include <memory>
#include "ntt.cuh"
#include <cstdio>
#include <cstring>
#include <cstdlib>

int gl_ctx::get_pow_table(gl_t base, gl_t scale, uint32_t hi_len, GlPowTable* out) {
    auto key = std::make_pair(base, scale);
    auto it = pow_tables.find(key);
    if (it != pow_tables.end() && it->second.hi_len >= hi_len) { *out = it->second; return GL_OK; }
    GlPowTable t;
    if (hi_len < 1) hi_len = 1;
    t.hi_len = hi_len;
    const uint32_t lo_len = 1u << NTT_SPLIT_LOG;
    GL_CHECK_HIP(hipMalloc((void**)&t.lo, (size_t)(lo_len + hi_len) * sizeof(gl_t)));
    t.hi = t.lo + lo_len;
    uint32_t m = lo_len > hi_len ? lo_len : hi_len;
    hipLaunchKernelGGL(ntt_power_table, dim3((m + 255) / 256), dim3(256), 0, stream, base, scale, t.lo, t.hi, hi_len);
    GL_CHECK_HIP(hipGetLastError());
    // an older, shorter table for the same key may still be in flight: it is parked until the context dies
    if (it != pow_tables.end()) retired_tables.push_back(it->second.lo);
    pow_tables[key] = t;
    *out = t;
    return GL_OK;
}

int gl_ctx::get_pass_table(gl_t w, gl_t scale, uint32_t lgN1, uint32_t lgN2, const gl_t** out) {
    auto key = std::make_tuple(w, scale, lgN1);
    auto it = pass_tables.find(key);
    if (it != pass_tables.end()) { *out = it->second; return GL_OK; }
    const size_t N = size_t(1) << (lgN1 + lgN2);
    gl_t* t = nullptr;
    GL_CHECK_HIP(hipMalloc((void**)&t, N * sizeof(gl_t)));
    hipLaunchKernelGGL(ntt_pass_table, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, w, scale, t, lgN1, lgN2);
    GL_CHECK_HIP(hipGetLastError());
    pass_tables[key] = t;
    *out = t;
    return GL_OK;
}

extern "C" int gl_ctx_create(int device, void* stream, gl_ctx** out) {
    GL_REQUIRE(out != nullptr, GL_ERR_ARG, "gl_ctx_create: out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return gl_fail(GL_ERR_HIP, "no HIP device available (this library has no CPU fallback)", __FILE__, __LINE__);
    GL_REQUIRE(device >= 0 && device < count, GL_ERR_ARG, "gl_ctx_create: bad device index");
    GL_CHECK_HIP(hipSetDevice(device));
    std::unique_ptr<gl_ctx, void (*)(gl_ctx*)> holder(new gl_ctx(), gl_ctx_release);      // nothing leaks on an error path
    gl_ctx* c = holder.get();
    c->device = device;
    if (const char* e = getenv("GL_NTT_SCRATCH_LOG")) { const int lg = atoi(e); if (lg >= 16 && lg <= 32) c->scratch_target = size_t(1) << lg; }      // tuning knob
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { GL_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    const uint32_t len = 1u << NTT_LOCAL_MAX_LOG;
    for (int dir = 0; dir < 2; dir++) {
        GL_CHECK_HIP(hipMalloc((void**)&c->tw_local[dir], len * sizeof(gl_t)));
        gl_t w = gl_host_root_of_unity(NTT_LOCAL_MAX_LOG);
        if (dir) w = gl_canon(gl_inv(w));
        hipLaunchKernelGGL(ntt_root_table, dim3(len / 256), dim3(256), 0, c->stream, w, c->tw_local[dir], len);
    }
    GL_CHECK_HIP(hipGetLastError());
    *out = holder.release();
    return GL_OK;
}
// ---------------------------------------------------------------------------------------- launchers
template <int LOGL, bool INV, bool ZP>
static int launch_col_impl(gl_ctx* c, const NttPassParams& p, dim3 grid) {
    constexpr size_t lds = NttGeom<LOGL>::LDS_BYTES;
    // once per instantiation and device; contexts on several host threads may get here together (setting it twice is harmless)
    static std::atomic<uint64_t> done{0};
    const uint64_t bit = uint64_t(1) << (c->device & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        GL_CHECK_HIP(hipFuncSetAttribute((const void*)ntt_col_pass<LOGL, INV, ZP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done.fetch_or(bit, std::memory_order_release);
    }
    GlTimed timed(c, INV ? "ntt_col_pass(inverse)" : "ntt_col_pass(forward)");
    hipLaunchKernelGGL((ntt_col_pass<LOGL, INV, ZP>), grid, dim3(NTT_THREADS), lds, c->stream, p);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
template <int LOGL, bool INV>
static int launch_col(gl_ctx* c, const NttPassParams& p, dim3 grid) {
    // zero-padded input (LDE with rate >= 8, forward only): the first radix stage sees one or two non-zero inputs per task
    if constexpr (!INV && LOGL >= 5 && LOGL <= 10 && ntt_first_radix(LOGL) <= 4) {
        const uint64_t N = uint64_t(1) << (p.lgN1 + p.lgN2);
        if ((uint64_t)p.n_in * 8 <= N) return launch_col_impl<LOGL, INV, true>(c, p, grid);
    }
    return launch_col_impl<LOGL, INV, false>(c, p, grid);
}
template <int LOGL, bool INV, bool SINGLE>
static int launch_row_impl(gl_ctx* c, const NttPassParams& p, dim3 grid) {
    constexpr size_t lds = NttGeom<LOGL>::LDS_BYTES;
    static std::atomic<uint64_t> done{0};
    const uint64_t bit = uint64_t(1) << (c->device & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        GL_CHECK_HIP(hipFuncSetAttribute((const void*)ntt_row_pass<LOGL, INV, SINGLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done.fetch_or(bit, std::memory_order_release);
    }
    GlTimed timed(c, INV ? "ntt_row_pass(inverse)" : "ntt_row_pass(forward)");
    hipLaunchKernelGGL((ntt_row_pass<LOGL, INV, SINGLE>), grid, dim3(NTT_THREADS), lds, c->stream, p);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
template <int LOGL, bool INV>
static int launch_row(gl_ctx* c, const NttPassParams& p, dim3 grid) {
    return p.lgN1 == 0 ? launch_row_impl<LOGL, INV, true>(c, p, grid) : launch_row_impl<LOGL, INV, false>(c, p, grid);
}

#define NTT_DISPATCH(fn, logl, inv, ...)                                             \
    switch (logl) {                                                                  \
        case 1: return (inv) ? fn<1, true>(__VA_ARGS__) : fn<1, false>(__VA_ARGS__);   \
        case 2: return (inv) ? fn<2, true>(__VA_ARGS__) : fn<2, false>(__VA_ARGS__);   \
        case 3: return (inv) ? fn<3, true>(__VA_ARGS__) : fn<3, false>(__VA_ARGS__);   \
        case 4: return (inv) ? fn<4, true>(__VA_ARGS__) : fn<4, false>(__VA_ARGS__);   \
        case 5: return (inv) ? fn<5, true>(__VA_ARGS__) : fn<5, false>(__VA_ARGS__);   \
        case 6: return (inv) ? fn<6, true>(__VA_ARGS__) : fn<6, false>(__VA_ARGS__);   \
        case 7: return (inv) ? fn<7, true>(__VA_ARGS__) : fn<7, false>(__VA_ARGS__);   \
        case 8: return (inv) ? fn<8, true>(__VA_ARGS__) : fn<8, false>(__VA_ARGS__);   \
        case 9: return (inv) ? fn<9, true>(__VA_ARGS__) : fn<9, false>(__VA_ARGS__);   \
        case 10: return (inv) ? fn<10, true>(__VA_ARGS__) : fn<10, false>(__VA_ARGS__); \
        case 11: return (inv) ? fn<11, true>(__VA_ARGS__) : fn<11, false>(__VA_ARGS__); \
        case 12: return (inv) ? fn<12, true>(__VA_ARGS__) : fn<12, false>(__VA_ARGS__); \
        default: return gl_fail(GL_ERR_ARG, "unsupported local NTT size", __FILE__, __LINE__); \
    }

static int dispatch_col(gl_ctx* c, int logl, bool inv, const NttPassParams& p, dim3 grid) { NTT_DISPATCH(launch_col, logl, inv, c, p, grid) }
static int dispatch_row(gl_ctx* c, int logl, bool inv, const NttPassParams& p, dim3 grid) { NTT_DISPATCH(launch_row, logl, inv, c, p, grid) }

__global__ void ntt_copy_canon(const gl_t* src, uint64_t src_stride, gl_t* dst, uint64_t dst_stride, uint32_t batch, gl_t f) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) dst[b * dst_stride] = gl_canon(gl_mul(src[b * src_stride], f));
}

int gl_ntt_run(gl_ctx* c, const gl_t* src, uint64_t src_stride, uint32_t n_in, gl_t* dst, uint64_t dst_stride,
               uint32_t lgN, uint32_t batch, bool inverse, gl_t pre_shift, gl_t post_shift, gl_t post_const) {
    GL_REQUIRE(c && src && dst, GL_ERR_ARG, "gl_ntt_run: null argument");
    GL_REQUIRE(lgN <= 2 * NTT_LOCAL_MAX_LOG && lgN <= 32, GL_ERR_ARG, "gl_ntt_run: transform too large");
    if (batch == 0) return GL_OK;
    GL_TRY(c->activate());
    const uint64_t N = uint64_t(1) << lgN;
    GL_REQUIRE(n_in >= 1 && n_in <= N, GL_ERR_ARG, "gl_ntt_run: bad input length");
    if (lgN == 0) {
        hipLaunchKernelGGL(ntt_copy_canon, dim3((batch + 255) / 256), dim3(256), 0, c->stream, src, src_stride, dst, dst_stride, batch, post_const);
        GL_CHECK_HIP(hipGetLastError());
        return GL_OK;
    }
    NttPassParams p;
    memset(&p, 0, sizeof p);
    p.tw_local = c->tw_local[inverse ? 1 : 0];
    p.post_const = 1;
#ifdef NTT_ABLATION
    { const char* dbg = getenv("GL_NTT_DEBUG"); p.debug = dbg ? (uint32_t)atoi(dbg) : 0; }
#endif
    GlPowTable pre, post, tw;
    if (pre_shift) GL_TRY(c->get_pow_table(gl_canon(pre_shift), 1, (n_in + (1u << NTT_SPLIT_LOG) - 1) >> NTT_SPLIT_LOG, &pre));
    if (post_shift) GL_TRY(c->get_pow_table(gl_canon(post_shift), gl_canon(post_const), (uint32_t)((N + (1u << NTT_SPLIT_LOG) - 1) >> NTT_SPLIT_LOG), &post));

    if (lgN <= NTT_LOCAL_MAX_LOG) {
        p.src = src; p.dst = dst; p.src_stride = src_stride; p.dst_stride = dst_stride;
        p.batch = batch; p.lgN1 = 0; p.lgN2 = lgN; p.n_in = n_in;
        if (pre_shift) { p.pre_lo = pre.lo; p.pre_hi = pre.hi; }
        if (post_shift) { p.post_lo = post.lo; p.post_hi = post.hi; } else p.post_const = gl_canon(post_const);
        const uint32_t T = 1u << (NTT_TILE_LOG - lgN);
        return dispatch_row(c, (int)lgN, inverse, p, dim3((batch + T - 1) / T, 1));
    }
    gl_t w = gl_host_root_of_unity(lgN);
    if (inverse) w = gl_canon(gl_inv(w));
    GL_TRY(c->get_pow_table(w, 1, (uint32_t)(N >> NTT_SPLIT_LOG), &tw));
    static_assert(NTT_NESTED_MIN_LOG >= 21, "the nested split needs M >= 2^13 (a full tile of columns in the rows' column pass); at 2^21 it measured slower than two passes (0.827 against 0.771 ms)");
    if (lgN >= NTT_NESTED_MIN_LOG) {
        // THREE passes, N = 2^10 x M: a column pass over the 2^10-point columns (stride M) with the twiddles w_N^(i2 k1), then the
        // M-point rows as a two-pass transform of their own whose row pass scatters output q of row k1 to k1 + 2^10 q.  Every pass has
        // columns of at most 2^10 points (tiles a wave owns); the 2^12-point tiles of the two-pass split run at 40 % of that rate.
        const uint32_t lgA = lgN >= 23 ? 10 : lgN - 13, lgM = lgN - lgA, lgM1 = lgM / 2, lgM2 = lgM - lgM1;
        const uint64_t M = uint64_t(1) << lgM;
        gl_t wm = gl_host_root_of_unity(lgM);
        if (inverse) wm = gl_canon(gl_inv(wm));
        GlPowTable twm;
        GL_TRY(c->get_pow_table(wm, 1, (uint32_t)(M >> NTT_SPLIT_LOG), &twm));
        gl_t row_post_const = gl_canon(post_const);
        const gl_t* twm_pass = nullptr;
        GL_TRY(c->get_pass_table(wm, post_shift ? gl_t(1) : row_post_const, lgM1, lgM2, &twm_pass));      // the scalar factor rides in the table
        if (!post_shift) row_post_const = 1;
        size_t want = c->scratch_target > N ? c->scratch_target : N;
        if (want > (size_t)batch * N) want = (size_t)batch * N;
        GL_TRY(c->ensure_scratch(want));
        uint32_t chunk_max = (uint32_t)(c->scratch_elems >> lgN);
        if (chunk_max > (65535u >> lgA)) chunk_max = 65535u >> lgA;                // the inner passes index rows through gridDim.y
        DevBuf second(c);
        GL_TRY(second.alloc(((size_t)(batch < chunk_max ? batch : chunk_max) << lgN) * sizeof(gl_t)));
        const uint32_t TA = 1u << (NTT_TILE_LOG - lgA), TB = 1u << (NTT_TILE_LOG - lgM1), TC = 1u << (NTT_TILE_LOG - lgM2);
        for (uint32_t b0 = 0; b0 < batch; b0 += chunk_max) {
            const uint32_t nb = (batch - b0) < chunk_max ? (batch - b0) : chunk_max;
            NttPassParams a = p;                                                    // columns of 2^10 points, stride M
            a.src = src + (uint64_t)b0 * src_stride; a.src_stride = src_stride;
            a.dst = c->scratch; a.dst_stride = N;
            a.batch = nb; a.lgN1 = lgA; a.lgN2 = lgM; a.n_in = n_in;
            a.tw_lo = tw.lo; a.tw_hi = tw.hi;
            if (pre_shift) { a.pre_lo = pre.lo; a.pre_hi = pre.hi; }
            GL_TRY(dispatch_col(c, (int)lgA, inverse, a, dim3((uint32_t)(M / TA), nb)));
            NttPassParams ic = p;                                                   // the rows (nb x 2^10 of them, contiguous): their column pass
            ic.src = c->scratch; ic.src_stride = M;
            ic.dst = second.as<gl_t>(); ic.dst_stride = M;
            ic.batch = nb << lgA; ic.lgN1 = lgM1; ic.lgN2 = lgM2; ic.n_in = (uint32_t)M;
            ic.tw_lo = twm.lo; ic.tw_hi = twm.hi; ic.tw_pass = twm_pass;
            GL_TRY(dispatch_col(c, (int)lgM1, inverse, ic, dim3((1u << lgM2) / TB, nb << lgA)));
            NttPassParams ir = p;                                                   // ... and their row pass, scattering into the result
            ir.src = second.as<gl_t>(); ir.src_stride = M;
            ir.dst = dst + (uint64_t)b0 * dst_stride; ir.dst_stride = dst_stride;
            ir.batch = nb << lgA; ir.lgN1 = lgM1; ir.lgN2 = lgM2; ir.n_in = (uint32_t)M; ir.out_shift = lgA;
            if (post_shift) { ir.post_lo = post.lo; ir.post_hi = post.hi; } else ir.post_const = row_post_const;
            GL_TRY(dispatch_row(c, (int)lgM2, inverse, ir, dim3(1u << lgM1, (nb << lgA) / TC)));      // a tile: one inner row of TC outer rows
        }
        return GL_OK;
    }
    const uint32_t lgN1 = lgN / 2, lgN2 = lgN - lgN1;
    // the N inter-pass twiddles as a table in output order (8 N bytes, shared by every polynomial of every batch): worth it
    // when several polynomials share it
    const gl_t* tw_pass = nullptr;
    gl_t row_post_const = gl_canon(post_const);
    if (lgN <= 22 && batch >= 4) {
        // a scalar output factor (the 1/N of an inverse transform) rides in the table: no multiply for it in the row pass
        const gl_t scale = post_shift ? gl_t(1) : row_post_const;
        GL_TRY(c->get_pass_table(w, scale, lgN1, lgN2, &tw_pass));
        if (!post_shift) row_post_const = 1;
    }
    size_t want = c->scratch_target > N ? c->scratch_target : N;
    if (want > (size_t)batch * N) want = (size_t)batch * N;
    GL_TRY(c->ensure_scratch(want));
    const uint32_t chunk_max = (uint32_t)(c->scratch_elems >> lgN);
    const uint32_t TA = 1u << (NTT_TILE_LOG - lgN1), TB = 1u << (NTT_TILE_LOG - lgN2);
    for (uint32_t b0 = 0; b0 < batch; b0 += chunk_max) {
        const uint32_t nb = (batch - b0) < chunk_max ? (batch - b0) : chunk_max;
        NttPassParams a = p;
        a.src = src + (uint64_t)b0 * src_stride; a.src_stride = src_stride;
        a.dst = c->scratch; a.dst_stride = N;
        a.batch = nb; a.lgN1 = lgN1; a.lgN2 = lgN2; a.n_in = n_in;
        a.tw_lo = tw.lo; a.tw_hi = tw.hi; a.tw_pass = tw_pass;
        if (pre_shift) { a.pre_lo = pre.lo; a.pre_hi = pre.hi; }
        GL_TRY(dispatch_col(c, (int)lgN1, inverse, a, dim3((1u << lgN2) / TA, nb)));
        NttPassParams r = p;
        r.src = c->scratch; r.src_stride = N;
        r.dst = dst + (uint64_t)b0 * dst_stride; r.dst_stride = dst_stride;
        r.batch = nb; r.lgN1 = lgN1; r.lgN2 = lgN2; r.n_in = (uint32_t)N;
        if (post_shift) { r.post_lo = post.lo; r.post_hi = post.hi; } else r.post_const = row_post_const;
        GL_TRY(dispatch_row(c, (int)lgN2, inverse, r, dim3((1u << lgN1) / TB, nb)));
    }
    return GL_OK;
}

// ------------------------------------------------------------------------------------- C ABI: NTTs
extern "C" int gl_ntt_forward(gl_ctx* c, uint64_t* d, uint32_t log_n, uint32_t batch) {
    return gl_ntt_run(c, d, uint64_t(1) << log_n, 1u << log_n, d, uint64_t(1) << log_n, log_n, batch, false, 0, 0, 1);
}
extern "C" int gl_ntt_inverse(gl_ctx* c, uint64_t* d, uint32_t log_n, uint32_t batch) {
    return gl_ntt_run(c, d, uint64_t(1) << log_n, 1u << log_n, d, uint64_t(1) << log_n, log_n, batch, true, 0, 0,
                      gl_host_inverse_2exp(log_n));
}
extern "C" int gl_ntt_coset_forward(gl_ctx* c, uint64_t* d, uint32_t log_n, uint32_t batch, uint64_t shift) {
    GL_REQUIRE(gl_canon(shift) != 0, GL_ERR_ARG, "coset shift must be non-zero");
    return gl_ntt_run(c, d, uint64_t(1) << log_n, 1u << log_n, d, uint64_t(1) << log_n, log_n, batch, false, shift, 0, 1);
}
extern "C" int gl_ntt_coset_inverse(gl_ctx* c, uint64_t* d, uint32_t log_n, uint32_t batch, uint64_t shift) {
    GL_REQUIRE(gl_canon(shift) != 0, GL_ERR_ARG, "coset shift must be non-zero");
    return gl_ntt_run(c, d, uint64_t(1) << log_n, 1u << log_n, d, uint64_t(1) << log_n, log_n, batch, true, 0,
                      gl_canon(gl_inv(shift)), gl_host_inverse_2exp(log_n));
}
extern "C" int gl_ntt_coset_lde(gl_ctx* c, const uint64_t* d_coeffs, uint32_t log_n, uint32_t rate_bits, uint32_t batch, uint64_t* d_out) {
    GL_REQUIRE(log_n + rate_bits <= 2 * NTT_LOCAL_MAX_LOG, GL_ERR_ARG, "LDE too large");
    return gl_ntt_run(c, d_coeffs, uint64_t(1) << log_n, 1u << log_n, d_out, uint64_t(1) << (log_n + rate_bits),
                      log_n + rate_bits, batch, false, GL_MULT_GENERATOR, 0, 1);
}
extern "C" int gl_fft_host(gl_ctx* c, uint64_t* h_data, uint32_t log_n, uint32_t batch, int inverse) {
    GL_REQUIRE(c && h_data, GL_ERR_ARG, "null argument");
    GL_TRY(c->activate());
    size_t bytes = ((size_t)batch << log_n) * sizeof(gl_t);
    gl_t* d = nullptr;
    GL_CHECK_HIP(hipMalloc((void**)&d, bytes ? bytes : 8));
    int st = gl_copy_h2d(c, d, h_data, bytes);
    if (st == GL_OK) st = inverse ? gl_ntt_inverse(c, d, log_n, batch) : gl_ntt_forward(c, d, log_n, batch);
    if (st == GL_OK) st = gl_copy_d2h(c, h_data, d, bytes);
    (void)hipFree(d);
    return st;
}

// ----------------------------------------------------------------------------- C ABI: field kernels
__global__ void k_field_op(int op, const gl_t* a, const gl_t* b, const gl_t* cc, gl_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl_t r = 0;
    switch (op) {
        case 0: r = gl_add(a[i], b[i]); break;
        case 1: r = gl_sub(a[i], b[i]); break;
        case 2: r = gl_mul(a[i], b[i]); break;
        case 3: r = gl_neg(a[i]); break;
        case 4: r = gl_inv(a[i]); break;
        case 5: r = a[i]; break;
        case 6: r = gl_mul_add(a[i], b[i], cc[i]); break;
        case 7: r = gl_mul_2exp(a[i], (unsigned)(b[i] % 192)); break;
    }
    out[i] = gl_canon(r);
}
__global__ void k_ext_op(int op, const gl_t* a, const gl_t* b, gl_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl2_t x = gl2_make(a[2 * i], a[2 * i + 1]), y = gl2_make(0, 0), r = gl2_make(0, 0);
    if (b) y = gl2_make(b[2 * i], b[2 * i + 1]);
    switch (op) {
        case 0: r = gl2_add(x, y); break;
        case 1: r = gl2_sub(x, y); break;
        case 2: r = gl2_mul(x, y); break;
        case 3: r = gl2_inv(x); break;
    }
    r = gl2_canon(r);
    out[2 * i] = r.a; out[2 * i + 1] = r.b;
}
extern "C" int gl_field_op(gl_ctx* c, int op, const uint64_t* a, const uint64_t* b, const uint64_t* cc, uint64_t* out, size_t n) {
    GL_REQUIRE(c && a && out && op >= 0 && op <= 7, GL_ERR_ARG, "gl_field_op: bad argument");
    GL_REQUIRE((op == 3 || op == 4 || op == 5) || b, GL_ERR_ARG, "gl_field_op: b is null");
    GL_REQUIRE(op != 6 || cc, GL_ERR_ARG, "gl_field_op: c is null");
    if (!n) return GL_OK;
    GL_TRY(c->activate());
    hipLaunchKernelGGL(k_field_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, a, b, cc, out, n);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
extern "C" int gl_ext_op(gl_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    GL_REQUIRE(c && a && out && op >= 0 && op <= 3, GL_ERR_ARG, "gl_ext_op: bad argument");
    GL_REQUIRE(op == 3 || b, GL_ERR_ARG, "gl_ext_op: b is null");
    if (!n) return GL_OK;
    GL_TRY(c->activate());
    hipLaunchKernelGGL(k_ext_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, a, b, out, n);
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
