// Batched Goldilocks NTT for gfx950: device kernels.
//
// Replaces the reference's per-column CPU radix-2 FFT (field/src/fft.rs:56-95,142-206) and the coset /
// zero-padding wrappers (field/src/polynomial/mod.rs:58-88,201-220,282-295).  Results are the same
// field elements (the DFT is unique); the algorithm is NOT the reference's: it is a two-pass
// ("four-step") decomposition N = N1*N2 so that every element crosses HBM twice per transform, each
// pass running a self-sorting Stockham FFT of up to 2^12 points inside LDS with radix-16/8 butterflies
// held in registers.  Butterfly twiddles inside a radix block are powers of two (the reference's 64th
// root of unity is 2^39), i.e. shifts; general twiddles are only applied between radix blocks (from a
// 4096-entry table) and once between the two passes (two-level table).
//
//   pass A ("column"): for T adjacent columns i2 of the N1 x N2 matrix x[i1*N2+i2]: FFT over i1,
//                      multiply by w_N^(i2*k1), store Y[k1*N2+i2]             (same footprint it read)
//   pass B ("row"):    for T adjacent rows k1: FFT over i2, store X[k1 + N1*k2] (T-element segments)
//   N <= 2^12:         a single row pass with N1 = 1 over T polynomials of the batch.
#pragma once
#include "gl64.cuh"

#define NTT_TILE_LOG 13                 // elements per workgroup tile (8192 * 8 B = 64 KiB of LDS)
#define NTT_THREADS 256
#define NTT_EPT ((1 << NTT_TILE_LOG) / NTT_THREADS)   // 32 elements per thread
#define NTT_LOCAL_MAX_LOG 12            // largest in-LDS transform
#define NTT_SPLIT_LOG 11                // two-level power tables: x^e = lo[e & 2047] * hi[e >> 11]

struct NttPassParams {
    const gl_t* src;
    gl_t* dst;
    uint64_t src_stride, dst_stride;   // elements between consecutive polynomials
    uint32_t batch;
    uint32_t lgN1, lgN2;               // N = N1 * N2 (single pass: lgN1 = 0)
    uint32_t n_in;                     // valid input elements per polynomial (rest read as zero)
    const gl_t* tw_local;              // w_4096^e (direction-specific), 4096 entries
    const gl_t* tw_lo; const gl_t* tw_hi;      // pass twiddle w_N^e, two-level (column pass)
    const gl_t* pre_lo; const gl_t* pre_hi;    // optional input scale by s^i (two-level), or null
    const gl_t* post_lo; const gl_t* post_hi;  // optional output scale by c*s^k (two-level), or null
    gl_t post_const;                   // scalar output factor when post tables are null (1 = none)
};

__host__ __device__ constexpr int ntt_first_radix(int rem) {
    return rem <= 4 ? rem : (rem == 5 || rem == 6 || rem == 9) ? 3 : 4;
}
__host__ __device__ constexpr unsigned ntt_bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// In-register DFT of R = 2^LOGR points (decimation in frequency); X[q] is left in u[bitrev(q)].
template <int LOGR, bool INV>
__device__ __forceinline__ void ntt_small_dft(gl_t* u) {
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int lg = LOGR; lg >= 1; lg--) {
        const int s = 1 << (lg - 1);
#pragma unroll
        for (int b = 0; b < R; b += 2 * s) {
#pragma unroll
            for (int i = 0; i < s; i++) {
                gl_t a = u[b + i], c = u[b + i + s];
                u[b + i] = gl_add(a, c);
                gl_t d = gl_sub(a, c);
                // w_{2s}^i = 2^(39 * (32/s) * i)
                unsigned e = (39u * (32u / (unsigned)s) * (unsigned)i) % 192u;
                if (INV) e = (192u - e) % 192u;
                u[b + i + s] = gl_mul_2exp(d, e);
            }
        }
    }
}

__device__ __forceinline__ gl_t ntt_pow2level(const gl_t* lo, const gl_t* hi, uint32_t e) {
    // hi[0] carries the table's scale factor, so the high part is always applied
    return gl_mul(lo[e & ((1u << NTT_SPLIT_LOG) - 1)], hi[e >> NTT_SPLIT_LOG]);
}

// One Stockham stage over the tile in LDS: L = 2^LOGL points per column, T = 2^LOGT columns,
// Ns = 2^LOGNS points already combined.  Layout: lds[i * LDT + t].
template <int LOGL, int LOGT, int LOGNS, bool INV>
__device__ __forceinline__ void ntt_lds_stages(gl_t* lds, const gl_t* __restrict__ tw_local, int tid) {
    if constexpr (LOGNS < LOGL) {
        constexpr int LOGR = ntt_first_radix(LOGL - LOGNS);
        constexpr int R = 1 << LOGR;
        constexpr int T = 1 << LOGT, LDT = T + 1;
        constexpr int TPT = NTT_EPT / R;                 // tasks per thread
        constexpr int LOGJ = LOGL - LOGR;                // tasks per column = 2^LOGJ
        constexpr int NS = 1 << LOGNS;
        gl_t u[TPT][R];
#pragma unroll
        for (int q = 0; q < TPT; q++) {
            const int task = tid + NTT_THREADS * q;
            const int t = task & (T - 1), j = task >> LOGT;
#pragma unroll
            for (int r = 0; r < R; r++) u[q][r] = lds[(j + (r << LOGJ)) * LDT + t];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPT; q++) {
            const int task = tid + NTT_THREADS * q;
            const int t = task & (T - 1), j = task >> LOGT;
            const int k = j & (NS - 1);
            if constexpr (LOGNS > 0) {
#pragma unroll
                for (int r = 1; r < R; r++)
                    u[q][r] = gl_mul(u[q][r], tw_local[(r * k) << (NTT_LOCAL_MAX_LOG - LOGNS - LOGR)]);
            }
            ntt_small_dft<LOGR, INV>(u[q]);
            const int j0 = ((j - k) << LOGR) + k;
#pragma unroll
            for (int o = 0; o < R; o++) lds[(j0 + (o << LOGNS)) * LDT + t] = u[q][ntt_bitrev(o, LOGR)];
        }
        __syncthreads();
        ntt_lds_stages<LOGL, LOGT, LOGNS + LOGR, INV>(lds, tw_local, tid);
    }
}

// COLUMN pass (pass A).  grid = (N2 / T, batch).
template <int LOGL, bool INV>
__global__ __launch_bounds__(NTT_THREADS) void ntt_col_pass(NttPassParams p) {
    constexpr int LOGT = NTT_TILE_LOG - LOGL, T = 1 << LOGT, LDT = T + 1, L = 1 << LOGL;
    extern __shared__ __align__(16) gl_t lds[];
    const int tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x << LOGT;
    const uint32_t b = blockIdx.y;
    const uint32_t lgN2 = p.lgN2;
    const gl_t* src = p.src + (uint64_t)b * p.src_stride;
    gl_t* dst = p.dst + (uint64_t)b * p.dst_stride;
    // load: e -> (i1 = e / T, t = e % T), zero beyond n_in
#pragma unroll 8
    for (int q = 0; q < NTT_EPT; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t t = e & (T - 1), i1 = e >> LOGT;
        const uint32_t i = (i1 << lgN2) + c0 + t;
        gl_t v = 0;
        if (i < p.n_in) {
            v = src[i];
            if (p.pre_lo) v = gl_mul(v, ntt_pow2level(p.pre_lo, p.pre_hi, i));
        }
        lds[i1 * LDT + t] = v;
    }
    __syncthreads();
    ntt_lds_stages<LOGL, LOGT, 0, INV>(lds, p.tw_local, tid);
    // store with the inter-pass twiddle w_N^(i2*k1)
#pragma unroll 4
    for (int q = 0; q < NTT_EPT; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t t = e & (T - 1), k1 = e >> LOGT;
        const uint32_t i2 = c0 + t;
        gl_t v = lds[k1 * LDT + t];
        const uint32_t ex = i2 * k1;
        if (ex) v = gl_mul(v, ntt_pow2level(p.tw_lo, p.tw_hi, ex));
        dst[((uint64_t)k1 << lgN2) + i2] = v;
    }
    (void)L;
}

// ROW pass (pass B, or the only pass when lgN1 == 0).
//   two-pass:   grid = (N1 / T, batch): rows k1 of one polynomial
//   single:     grid = (ceil(batch / T), 1): T polynomials
template <int LOGL, bool INV>
__global__ __launch_bounds__(NTT_THREADS) void ntt_row_pass(NttPassParams p) {
    constexpr int LOGT = NTT_TILE_LOG - LOGL, T = 1 << LOGT, LDT = T + 1, L = 1 << LOGL;
    extern __shared__ __align__(16) gl_t lds[];
    const int tid = threadIdx.x;
    const bool single = (p.lgN1 == 0);
    const uint32_t r0 = blockIdx.x << LOGT;                 // first row (k1) or first polynomial
    const uint32_t b = single ? 0 : blockIdx.y;
    const gl_t* src = p.src + (uint64_t)b * p.src_stride;
    gl_t* dst = p.dst + (uint64_t)b * p.dst_stride;
    // load: e -> (r = e / L, i2 = e % L): contiguous rows
#pragma unroll 8
    for (int q = 0; q < NTT_EPT; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t i2 = e & (L - 1), r = e >> LOGL;
        gl_t v = 0;
        if (single) {
            const uint32_t poly = r0 + r;
            if (poly < p.batch && i2 < p.n_in) {
                v = p.src[(uint64_t)poly * p.src_stride + i2];
                if (p.pre_lo) v = gl_mul(v, ntt_pow2level(p.pre_lo, p.pre_hi, i2));
            }
        } else {
            v = src[((uint64_t)(r0 + r) << LOGL) + i2];
        }
        lds[i2 * LDT + r] = v;
    }
    __syncthreads();
    ntt_lds_stages<LOGL, LOGT, 0, INV>(lds, p.tw_local, tid);
    if (single) {
        // store rows contiguously: e -> (r = e / L, k = e % L)
#pragma unroll 4
        for (int q = 0; q < NTT_EPT; q++) {
            const int e = tid + NTT_THREADS * q;
            const uint32_t k = e & (L - 1), r = e >> LOGL;
            const uint32_t poly = r0 + r;
            if (poly < p.batch) {
                gl_t v = lds[k * LDT + r];
                if (p.post_lo) v = gl_mul(v, ntt_pow2level(p.post_lo, p.post_hi, k));
                else if (p.post_const != 1) v = gl_mul(v, p.post_const);
                p.dst[(uint64_t)poly * p.dst_stride + k] = gl_canon(v);
            }
        }
    } else {
        const uint32_t lgN1 = p.lgN1;
        // store X[k1 + N1*k2]: e -> (k2 = e / T, r = e % T): T-element segments
#pragma unroll 4
        for (int q = 0; q < NTT_EPT; q++) {
            const int e = tid + NTT_THREADS * q;
            const uint32_t r = e & (T - 1), k2 = e >> LOGT;
            const uint32_t k = (r0 + r) + (k2 << lgN1);
            gl_t v = lds[k2 * LDT + r];
            if (p.post_lo) v = gl_mul(v, ntt_pow2level(p.post_lo, p.post_hi, k));
            else if (p.post_const != 1) v = gl_mul(v, p.post_const);
            dst[k] = gl_canon(v);
        }
    }
}

// out_lo[j] = base^j (j < 2^SPLIT), out_hi[j] = scale * base^(j << SPLIT) (j < hi_len)
__global__ void ntt_power_table(gl_t base, gl_t scale, gl_t* out_lo, gl_t* out_hi, uint32_t hi_len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo_len = 1u << NTT_SPLIT_LOG;
    if (j < lo_len) out_lo[j] = gl_canon(gl_exp(base, j));
    if (j < hi_len) out_hi[j] = gl_canon(gl_mul(scale, gl_exp(base, (uint64_t)j << NTT_SPLIT_LOG)));
}
__global__ void ntt_root_table(gl_t base, gl_t* out, uint32_t len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < len) out[j] = gl_canon(gl_exp(base, j));
}
