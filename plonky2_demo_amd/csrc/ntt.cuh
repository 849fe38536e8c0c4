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
//
// Inside a pass every value is kept CANONICAL (< p): inputs are canonicalised when loaded, and add / subtract / shift-twiddle /
// multiply each return the canonical representative with ONE fix-up (gl64_gfx950.cuh: 5 instructions for an add or a subtract
// instead of 7 / 10 with two fix-ups, 6-11 for a shift twiddle instead of 8-26; general multiplies three at a time).  The stores
// need no final canonicalisation.
#pragma once
#include <type_traits>
#include "gl64.cuh"
#include "gl64_gfx950.cuh"

#ifndef NTT_TILE_LOG
#define NTT_TILE_LOG 13                 // elements per workgroup tile (8192 * 8 B = 64 KiB of LDS)
#endif
#ifndef NTT_THREADS
#define NTT_THREADS 512                 // 512: 16 elements per thread, radix-16 register blocks, 4 waves per SIMD (2 workgroups per CU);
#endif                                  // 1024: 8 per thread, radix-8 blocks, 8 waves per SIMD when the kernel fits 64 VGPRs
#define NTT_EPT ((1 << NTT_TILE_LOG) / NTT_THREADS)   // elements per thread
#ifndef NTT_WAVES_PER_SIMD
#define NTT_WAVES_PER_SIMD (NTT_THREADS / 64 / 4 * 2)  // two workgroups per CU (LDS)
#endif
#define NTT_LOCAL_MAX_LOG 12            // largest in-LDS transform
#ifndef NTT_NESTED_MIN_LOG
#define NTT_NESTED_MIN_LOG 22           // from 2^22 points on: three passes (2^9 or 2^10-point columns x a two-pass M >= 2^13) instead of two
                                        // with 2^11 / 2^12-point tiles (M >= 2^13: the rows' column pass needs a full tile of columns)
#endif
#ifndef NTT_COL_DIRECT
#define NTT_COL_DIRECT 1
#endif
#ifndef NTT_TWPASS_HALVES
#define NTT_TWPASS_HALVES 1
#endif
#ifdef NTT_ABLATION
#define NTT_DBG(p, bit) ((p).debug & (bit))
#define NTT_ABLATION_BUILD 1            // the diagnostic build keeps the staged loads, whose pieces it can switch off
#else
#define NTT_DBG(p, bit) 0
#define NTT_ABLATION_BUILD 0
#endif
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
    const gl_t* tw_pass;                       // or, when non-null, the table w_N^(i2*k1) laid out like the column pass's
                                               // output ([k1][i2]): one coalesced load instead of two lookups and a multiply
    const gl_t* pre_lo; const gl_t* pre_hi;    // optional input scale by s^i (two-level), or null
    const gl_t* post_lo; const gl_t* post_hi;  // optional output scale by c*s^k (two-level), or null
    gl_t post_const;                   // scalar output factor when post tables are null (1 = none)
    uint32_t out_shift;                // row pass of a NESTED transform (N = 2^out_shift x M, this pass finishes the M-point rows): batch entry
                                       // b is outer row (b & (2^out_shift - 1)) of polynomial b >> out_shift and output k of it goes to
                                       // (k << out_shift) + outer row.  A tile then holds ONE inner row (blockIdx.x) of T consecutive batch
                                       // entries (blockIdx.y * T ..), so that its scattered stores are T-element segments as well.
#ifdef NTT_ABLATION
    uint32_t debug;                    // diagnostic builds only (-DNTT_ABLATION, env GL_NTT_DEBUG): 1 skip stages, 2 skip loads, 4 skip stores
#endif
};

__host__ __device__ constexpr int ntt_first_radix(int rem) {
    if (NTT_EPT >= 32) return rem <= 5 ? rem : (rem == 6) ? 3 : (rem == 7 || rem == 8) ? 4 : 5;     // radix at most 32: 10 = 5 + 5
    // 2^10 = 16 x 16 x 4 (round 3, late; 16 x 8 x 8 before, -DNTT_PLAN_16_8_8): 3/4 instead of 7/8 of the last stage's inputs carry a
    // general twiddle and a task loads 3 instead of 7 of them: 0.577 -> 0.566 ms for the 2^20 x 64 transform
#ifdef NTT_PLAN_16_8_8
    if (NTT_EPT >= 16) return rem <= 4 ? rem : (rem == 5 || rem == 6 || rem == 9) ? 3 : 4;
#endif
    if (NTT_EPT >= 16) return rem <= 4 ? rem : (rem == 5) ? 3 : 4;         // 2^9 = 16 x 8 x 4 (8 x 16 x 4 before: 0.1445 -> 0.1409 ms for 2^18 x 64)
    return rem <= 3 ? rem : (rem == 4 ? 2 : 3);          // radix at most 8: 10 = 3 + 3 + 2 + 2
}
__host__ __device__ constexpr unsigned ntt_bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// x * 2^e for a compile-time e in [0, 96), canonical in and out
template <unsigned E>
__device__ __forceinline__ gl_t ntt_shift_twiddle(gl_t x) {
    if constexpr (E == 0) return x;
    else return glx_shl_c<(int)E>(x);
}
// exponent of the twiddle of butterfly K (0 <= K < R/2) of the layer with half-size s = 2^(LG-1): butterfly K pairs u[b + i] with
// u[b + i + s], b = (K / s) * 2s, i = K % s, twiddle w_{2s}^i = 2^(39 * (32/s) * i) (the reference's w_64 = 2^39)
template <int LOGR, bool INV, int LG, int K>
struct NttBfly {
    static constexpr int s = 1 << (LG - 1), i = K % s, lo = (K / s) * 2 * s + i, hi = lo + s;
    static constexpr unsigned e0 = (39u * (32u / (unsigned)s) * (unsigned)i) % 192u;
    static constexpr unsigned e = INV ? (192u - e0) % 192u : e0;
};
// differences of the butterflies K0 .. K0+3 of a layer, four at a time (2^96 = -1: a twiddle -2^k is applied as (c - a) * 2^k)
template <int LOGR, bool INV, int LG, int K0>
__device__ __forceinline__ void ntt_layer_diffs4(const gl_t* u, gl_t* d) {
    gl_t a[4], b[4], r[4];
#define NTT_PICK(k)                                                                                   \
    { using Bf = NttBfly<LOGR, INV, LG, K0 + k>;                                                      \
      if constexpr (Bf::e >= 96u) { a[k] = u[Bf::hi]; b[k] = u[Bf::lo]; } else { a[k] = u[Bf::lo]; b[k] = u[Bf::hi]; } }
    NTT_PICK(0) NTT_PICK(1) NTT_PICK(2) NTT_PICK(3)
#undef NTT_PICK
    glx_sub_cc4(a, b, r);
    d[K0] = r[0]; d[K0 + 1] = r[1]; d[K0 + 2] = r[2]; d[K0 + 3] = r[3];
}
template <int LOGR, bool INV, int LG, int K>
__device__ __forceinline__ void ntt_layer_finish(gl_t* u, const gl_t* d) {
    constexpr int H = 1 << (LOGR - 1);
    if constexpr (K < H) {
        using Bf = NttBfly<LOGR, INV, LG, K>;
        u[Bf::lo] = glx_add_cc(u[Bf::lo], u[Bf::hi]);
        u[Bf::hi] = ntt_shift_twiddle<Bf::e % 96u>(d[K]);
        ntt_layer_finish<LOGR, INV, LG, K + 1>(u, d);
    }
}
template <int LOGR, bool INV, int LG>
__device__ __forceinline__ void ntt_small_dft_l(gl_t* u) {
    if constexpr (LG >= 1) {
        constexpr int H = 1 << (LOGR - 1);              // butterflies per layer
        gl_t d[H];
        if constexpr (H >= 4) {
            ntt_layer_diffs4<LOGR, INV, LG, 0>(u, d);
            if constexpr (H >= 8) ntt_layer_diffs4<LOGR, INV, LG, 4>(u, d);
            if constexpr (H >= 16) { ntt_layer_diffs4<LOGR, INV, LG, 8>(u, d); ntt_layer_diffs4<LOGR, INV, LG, 12>(u, d); }
            static_assert(H <= 16, "radix at most 32");
        } else {
#pragma unroll
            for (int k = 0; k < H; k++) {
                // H is 1 or 2: single subtractions
                const int s = 1 << (LG - 1), i = k % s, lo = (k / s) * 2 * s + i, hi = lo + s;
                unsigned e = (39u * (32u / (unsigned)s) * (unsigned)i) % 192u;
                if (INV) e = (192u - e) % 192u;
                d[k] = (e >= 96u) ? glx_sub_cc(u[hi], u[lo]) : glx_sub_cc(u[lo], u[hi]);
            }
        }
        ntt_layer_finish<LOGR, INV, LG, 0>(u, d);
        ntt_small_dft_l<LOGR, INV, LG - 1>(u);
    }
}
// In-register DFT of R = 2^LOGR points (decimation in frequency) on canonical values; X[q] is left in u[bitrev(q)].
template <int LOGR, bool INV>
__device__ __forceinline__ void ntt_small_dft(gl_t* u) { ntt_small_dft_l<LOGR, INV, LOGR>(u); }

// v[k] *= tw[k] for k < COUNT, canonical results: products go three at a time (glx_mul3)
template <int COUNT>
__device__ __forceinline__ void ntt_mul_many(gl_t* v, const gl_t* tw) {
    constexpr int G = COUNT / 3 * 3;
#pragma unroll
    for (int k = 0; k < G; k += 3) glx_mul3<true>(v[k], tw[k], v[k + 1], tw[k + 1], v[k + 2], tw[k + 2], v[k], v[k + 1], v[k + 2]);
#pragma unroll
    for (int k = G; k < COUNT; k++) v[k] = glx_mul<true>(v[k], tw[k]);
}

__device__ __forceinline__ gl_t ntt_pow2level(const gl_t* lo, const gl_t* hi, uint32_t e) {
    // hi[0] carries the table's scale factor, so the high part is always applied
    return glx_mul<true>(lo[e & ((1u << NTT_SPLIT_LOG) - 1)], hi[e >> NTT_SPLIT_LOG]);
}

// ---- LDS tile layout --------------------------------------------------------------------------------
// LOGL <= 10 ("wave-owned"): the tile holds T columns of L points, column-major: lds[t * LW + phi(i)], phi(i) = i + i/16.
// One wave (64 lanes x 16 elements) owns 1024/L whole columns, so all radix stages of a column are
// wave-local: they need no workgroup barrier, only program order (LDS executes a wave's accesses in
// order) and a compiler-level wave barrier.  phi() pads one element per 16 so that the stride-16 writes of the first
// radix-16 stage hit 32 different bank pairs (stride 17), and -- unlike an XOR swizzle -- it is additive over the index
// pieces a stage uses (task base + r * 2^k with k >= 4, or 16 j + o): every LDS access of an unrolled stage is
// "one base register + immediate offset", no per-element address arithmetic.
// LOGL > 10: columns span several waves: layout lds[i * (T+1) + t] with workgroup barriers per stage.
template <int LOGL>
struct NttGeom {
    static constexpr int LOGT = NTT_TILE_LOG - LOGL, T = 1 << LOGT, L = 1 << LOGL;
    static constexpr bool WAVE_OWNED = ((1 << LOGL) <= 64 * NTT_EPT);          // a wave (64 lanes x NTT_EPT elements) holds whole columns
    static constexpr int LW = L + (L >> 4) + (T >= 32 ? 1 : 32 / T);           // column stride (elements), see at()
    static constexpr int LDT = T + 1;                                          // row stride of the legacy layout
#ifdef NTT_LDS_ALIAS    // DIAGNOSTIC ONLY (wrong results): the tile aliased onto 32 KiB, to time a higher occupancy before building it
    static constexpr size_t LDS_BYTES = 32768 + (WAVE_OWNED ? (1 << LOGL) * 8 : 0);
    static constexpr int TW_OFF = 4096;
    __device__ static __forceinline__ int at(int t, int i) {
        if constexpr (WAVE_OWNED) return (t * LW + i + (i >> 4)) & 4095;
        else return (i * LDT + t) & 4095;
    }
#else
    // wave-owned tiles also keep the stage twiddles w_L^e (e < L, 8 L bytes) behind the tile: 70 + 8 KiB for L = 1024, two
    // workgroups per CU still fit the 160 KiB.  Loaded from global memory the 7 or 15 twiddles of a task were the largest
    // single stall of a pass (round 3: 0.713 -> 0.634 ms for the 2^20 x 64 transform with the loads taken out).
    static constexpr int TW_OFF = WAVE_OWNED ? T * LW : 0;                     // element offset of the table
    static constexpr size_t LDS_BYTES = WAVE_OWNED ? ((size_t)T * LW + L) * 8 : (size_t)L * LDT * 8;
    __device__ static __forceinline__ int at(int t, int i) {
        if constexpr (WAVE_OWNED) return t * LW + i + (i >> 4);
        else return i * LDT + t;
    }
#endif
};

__device__ __forceinline__ void ntt_wave_sync() {
    // orders this wave's LDS accesses (hardware keeps them in issue order); stops compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// zero-padded first stage (radix 16 or 8): output o of the task is x0 + w_R^o x1 (radix 8: x0), canonical
template <int R, bool INV, int O>
__device__ __forceinline__ void ntt_zp_first_all(gl_t x0, gl_t x1, gl_t* out) {
    if constexpr (O < R) {
        if constexpr (R != 16) out[O] = x0;
        else if constexpr (O == 0) out[O] = glx_add_cc(x0, x1);
        else {              // w_16^o = 2^(156 o) (the reference's w_64 = 2^39); 2^96 = -1
            constexpr unsigned e0 = (156u * (unsigned)O) % 192u;
            constexpr unsigned e = INV ? (192u - e0) % 192u : e0;
            const gl_t tt = ntt_shift_twiddle<e % 96u>(x1);
            out[O] = (e >= 96u) ? glx_sub_cc(x0, tt) : glx_add_cc(x0, tt);
        }
        ntt_zp_first_all<R, INV, O + 1>(x0, x1, out);
    }
}

// All Stockham stages over the tile in LDS: L = 2^LOGL points per column, Ns = 2^LOGNS already combined.
// ZP ("zero padded", column pass of an LDE with rate >= 8): only the first L/8 points of every column are non-zero, so
// of the R inputs i = j + r L/R of a first-stage task only r < R/8 exist: a radix-8 first stage is a broadcast, a radix-16
// one is x0 + w_16^o x1.  The zero points are neither stored to nor read from LDS.
template <int LOGL, int LOGNS, bool INV, bool ZP = false>
__device__ __forceinline__ void ntt_lds_stages(gl_t* lds, const gl_t* __restrict__ tw_local, int tid) {
    if constexpr (LOGNS < LOGL) {
        using G = NttGeom<LOGL>;
        constexpr int LOGR = ntt_first_radix(LOGL - LOGNS);
        constexpr int R = 1 << LOGR;
        constexpr int TPT = NTT_EPT / R;                 // tasks per thread
        constexpr int LOGJ = LOGL - LOGR;                // tasks per column = 2^LOGJ
        constexpr int NS = 1 << LOGNS;
        gl_t u[TPT][R];
        int tcol[TPT], tj[TPT];
#pragma unroll
        for (int q = 0; q < TPT; q++) {
            if constexpr (G::WAVE_OWNED) {
                constexpr int CPW = G::T / (NTT_THREADS / 64);       // columns per wave
                const int wave = tid >> 6, lane = tid & 63;
                const int w_task = lane + 64 * q;                    // < 1024 / R
                tcol[q] = wave * CPW + (w_task >> LOGJ);
                tj[q] = w_task & ((1 << LOGJ) - 1);
            } else {
                const int task = tid + NTT_THREADS * q;
                tcol[q] = task & (G::T - 1);
                tj[q] = task >> G::LOGT;
            }
            constexpr int RIN = (ZP && LOGNS == 0) ? R / 8 : R;
#pragma unroll
            for (int r = 0; r < RIN; r++) u[q][r] = lds[G::at(tcol[q], tj[q] + (r << LOGJ))];
        }
        if constexpr (G::WAVE_OWNED) ntt_wave_sync(); else __syncthreads();
        if constexpr (ZP && LOGNS == 0) {
            static_assert(!ZP || LOGR >= 3, "zero-padded first stage needs radix >= 8");
#pragma unroll
            for (int q = 0; q < TPT; q++) {
                const int j0 = tj[q] << LOGR;
                gl_t outv[R];
                ntt_zp_first_all<R, INV, 0>(u[q][0], u[q][1], outv);
#pragma unroll
                for (int o = 0; o < R; o++) lds[G::at(tcol[q], j0 + o)] = outv[o];
            }
        } else
#pragma unroll
        for (int q = 0; q < TPT; q++) {
            const int j = tj[q];
            const int k = j & (NS - 1);
            if constexpr (LOGNS > 0) {
                gl_t tw[R - 1];
#pragma unroll
#ifdef NTT_NO_TW_LOADS      // DIAGNOSTIC ONLY (wrong results): what the stage twiddle loads from global memory cost
                for (int r = 1; r < R; r++) tw[r - 1] = (gl_t)(r * k + 1) * 0x9E3779B97F4A7C15ULL >> 1;
#else
                for (int r = 1; r < R; r++) {
                    // wave-owned tiles: from the table in LDS (filled by ntt_fill_lds_twiddles before the first barrier)
                    if constexpr (G::WAVE_OWNED) tw[r - 1] = lds[G::TW_OFF + ((r * k) << (LOGL - LOGNS - LOGR))];
                    else tw[r - 1] = tw_local[(r * k) << (NTT_LOCAL_MAX_LOG - LOGNS - LOGR)];
                }
#endif
                ntt_mul_many<R - 1>(&u[q][1], tw);
            }
            ntt_small_dft<LOGR, INV>(u[q]);
            const int j0 = ((j - k) << LOGR) + k;
#pragma unroll
            for (int o = 0; o < R; o++) lds[G::at(tcol[q], j0 + (o << LOGNS))] = u[q][ntt_bitrev(o, LOGR)];
        }
        if constexpr (G::WAVE_OWNED) ntt_wave_sync(); else __syncthreads();
        ntt_lds_stages<LOGL, LOGNS + LOGR, INV, ZP>(lds, tw_local, tid);
    }
}

// First Stockham stage of a wave-owned tile with its inputs taken straight from global memory (`load(t, i)` = point i of column
// t of the tile as stored, `fix(t, i, v)` = its canonical, pre-scaled value: two steps, so that all loads are in flight together), instead of from a tile that was first staged through LDS: one LDS round trip and -- in the row pass,
// where a wave reads its own rows -- the workgroup barrier in front of the first stage go away.  CROSS: tasks are dealt so that
// consecutive lanes take consecutive COLUMNS (the column pass: a row of T columns is one contiguous global segment); otherwise
// consecutive lanes take consecutive points of one column (the row pass: rows are contiguous).  The outputs land at their
// Stockham positions; the caller synchronises (workgroup barrier: the twiddle table, and with CROSS the columns, were written by
// other waves) and continues with ntt_lds_stages<LOGL, first radix>.
template <int LOGL, bool INV, bool ZP, bool CROSS, class LOAD, class FIX, class MID>
__device__ __forceinline__ void ntt_first_stage_direct(gl_t* lds, int tid, LOAD load, FIX fix, MID while_loading) {
    using G = NttGeom<LOGL>;
    static_assert(G::WAVE_OWNED, "direct first stage: wave-owned tiles only");
    constexpr int LOGR = ntt_first_radix(LOGL);
    constexpr int R = 1 << LOGR;
    constexpr int TPT = NTT_EPT / R;
    constexpr int LOGJ = LOGL - LOGR;
    constexpr int RIN = ZP ? R / 8 : R;
    static_assert(!ZP || LOGR >= 3, "zero-padded first stage needs radix >= 8");
    gl_t u[TPT][R];
    int tcol[TPT], tj[TPT];
#pragma unroll
    for (int q = 0; q < TPT; q++) {
        if constexpr (CROSS) {
            const int task = tid + NTT_THREADS * q;
            tcol[q] = task & (G::T - 1);
            tj[q] = task >> G::LOGT;
        } else {
            constexpr int CPW = G::T / (NTT_THREADS / 64);
            const int wave = tid >> 6, lane = tid & 63;
            const int w_task = lane + 64 * q;
            tcol[q] = wave * CPW + (w_task >> LOGJ);
            tj[q] = w_task & ((1 << LOGJ) - 1);
        }
#pragma unroll
        for (int r = 0; r < RIN; r++) u[q][r] = load(tcol[q], tj[q] + (r << LOGJ));     // every request goes out before the first use
    }
    while_loading();                // work that does not need the inputs, placed under their memory latency
#pragma unroll
    for (int q = 0; q < TPT; q++) {
#pragma unroll
        for (int r = 0; r < RIN; r++) u[q][r] = fix(tcol[q], tj[q] + (r << LOGJ), u[q][r]);
    }
#pragma unroll
    for (int q = 0; q < TPT; q++) {
        const int j0 = tj[q] << LOGR;
        if constexpr (ZP) {
            gl_t outv[R];
            ntt_zp_first_all<R, INV, 0>(u[q][0], u[q][1], outv);
#pragma unroll
            for (int o = 0; o < R; o++) lds[G::at(tcol[q], j0 + o)] = outv[o];
        } else {
            ntt_small_dft<LOGR, INV>(u[q]);
#pragma unroll
            for (int o = 0; o < R; o++) lds[G::at(tcol[q], j0 + o)] = u[q][ntt_bitrev(o, LOGR)];
        }
    }
}

// w_L^e, e < L, into the LDS table behind a wave-owned tile (visible after the workgroup barrier that follows the tile load)
template <int LOGL>
__device__ __forceinline__ void ntt_fill_lds_twiddles(gl_t* lds, const gl_t* __restrict__ tw_local, int tid) {
    using G = NttGeom<LOGL>;
    if constexpr (G::WAVE_OWNED && LOGL >= 4) {         // transforms of <= 2^4 points are one radix block: no stage twiddles
#pragma unroll
        for (int e = tid; e < G::L; e += NTT_THREADS) lds[G::TW_OFF + e] = tw_local[e << (NTT_LOCAL_MAX_LOG - LOGL)];
    }
}

// Element `i` of a polynomial through a 32-bit BYTE offset from the (uniform) base: the access becomes "scalar base + one 32-bit
// vector offset" instead of four 64-bit address instructions per element.  Valid for i < 2^29 (transforms are at most 2^24).
__device__ __forceinline__ gl_t ntt_ld(const gl_t* base, uint32_t i) { return *(const gl_t*)((const char*)base + (i << 3)); }
__device__ __forceinline__ void ntt_st(gl_t* base, uint32_t i, gl_t x) { *(gl_t*)((char*)base + (i << 3)) = x; }

// One workgroup per tile, deliberately NOT persistent: workgroups that walk several tiles (grid = what is resident, stage twiddles
// staged once, stores draining under the next tile's loads) were built and measured in round 3 at 0.74 ms against 0.63 ms for the
// 2^20 x 64 transform -- started together and doing equal work, the two workgroups of a CU stay in phase (both in their memory
// phase, then both in their butterflies); the hardware dispatcher's one-in-one-out replacement keeps them apart, a start-up
// stagger of half a tile does not (profiles/README.md).

// COLUMN pass (pass A).  grid = (N2 / T, batch): T adjacent columns of one polynomial per workgroup.
template <int LOGL, bool INV, bool ZP>
__device__ __forceinline__ void ntt_col_tile(const NttPassParams& p, gl_t* lds, const int tid, const uint32_t bx, const uint32_t b) {
    using G = NttGeom<LOGL>;
    constexpr int LOGT = G::LOGT, T = G::T;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so give
    // XCD x a contiguous run of column tiles: neighbouring tiles share 128-byte lines when T*8 < 128.
    uint32_t tile = bx;
    if ((gridDim.x & 7) == 0) tile = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const uint32_t c0 = tile << LOGT;
    const uint32_t lgN2 = p.lgN2;
    const gl_t* src = p.src + (uint64_t)b * p.src_stride;
    gl_t* dst = p.dst + (uint64_t)b * p.dst_stride;
    if constexpr (G::WAVE_OWNED && !NTT_ABLATION_BUILD && NTT_COL_DIRECT) {
        // wave-owned tiles: the first radix stage reads its inputs straight from global memory (lanes across the T columns)
        ntt_fill_lds_twiddles<LOGL>(lds, p.tw_local, tid);
        const uint32_t n_in = p.n_in;
        const gl_t* pre_lo = p.pre_lo; const gl_t* pre_hi = p.pre_hi;
        // FULL: every point the tile reads exists (wave-uniform; always so for a transform of n_in = N points): no per-element
        // range check, i.e. no exec-masked region and no zero fill around each of the thread's 16 loads
        auto first_stage = [&](auto full) {
            constexpr bool FULL = decltype(full)::value;
            ntt_first_stage_direct<LOGL, INV, ZP, true>(lds, tid,
                [&](int t, int i1) -> gl_t {
                    const uint32_t i = ((uint32_t)i1 << lgN2) + c0 + (uint32_t)t;
                    if constexpr (FULL) return ntt_ld(src, i);
                    else return (i < n_in) ? ntt_ld(src, i) : (gl_t)0;
                },
                [&](int t, int i1, gl_t v) -> gl_t {
                    const uint32_t i = ((uint32_t)i1 << lgN2) + c0 + (uint32_t)t;
                    if (pre_lo && (FULL || i < n_in)) return glx_mul<true>(v, ntt_pow2level(pre_lo, pre_hi, i));
                    return glx_canon(v);                  // the caller's values may be any u64 representatives
                },
                [] {});
        };
        constexpr uint32_t LAST_ROW = ZP ? (G::L / 8 - 1) : (G::L - 1);          // the last row of the tile that is read at all
        if (((LAST_ROW << lgN2) + c0 + (uint32_t)T - 1) < n_in) first_stage(std::true_type{}); else first_stage(std::false_type{});
        __syncthreads();
        ntt_lds_stages<LOGL, ntt_first_radix(LOGL), INV, ZP>(lds, p.tw_local, tid);
    } else {
    // load: e -> (i1 = e / T, t = e % T), zero beyond n_in; with ZP rows >= L/8 are known zeros and never touched
    constexpr int QLOAD = ZP ? NTT_EPT / 8 : NTT_EPT;
    gl_t v[QLOAD];
#pragma unroll
    for (int q = 0; q < QLOAD; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t t = e & (T - 1), i1 = e >> LOGT;
        const uint32_t i = (i1 << lgN2) + c0 + t;
        v[q] = NTT_DBG(p, 2) ? (gl_t)e : ((i < p.n_in) ? ntt_ld(src, i) : (gl_t)0);
    }
#pragma unroll
    for (int q = 0; q < QLOAD; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t t = e & (T - 1), i1 = e >> LOGT;
        const uint32_t i = (i1 << lgN2) + c0 + t;
        gl_t x;
        if (p.pre_lo && i < p.n_in) x = glx_mul<true>(v[q], ntt_pow2level(p.pre_lo, p.pre_hi, i));
        else x = glx_canon(v[q]);             // the caller's values may be any u64 representatives
        lds[G::at(t, i1)] = x;
    }
    ntt_fill_lds_twiddles<LOGL>(lds, p.tw_local, tid);
    __syncthreads();
    if (!NTT_DBG(p, 1)) ntt_lds_stages<LOGL, 0, INV, ZP>(lds, p.tw_local, tid);
    }
    if constexpr (G::WAVE_OWNED) __syncthreads();
    // store with the inter-pass twiddle w_N^(i2*k1); the products go three at a time
    const gl_t* tw_pass = p.tw_pass;
    auto elem = [&](int q, gl_t& x, gl_t& tw, uint32_t& o) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t t = e & (T - 1), k1 = e >> LOGT;
        const uint32_t i2 = c0 + t;
        x = lds[G::at(t, k1)];
        o = (k1 << lgN2) + i2;
        tw = tw_pass ? ntt_ld(tw_pass, o) : ntt_pow2level(p.tw_lo, p.tw_hi, i2 * k1);
    };
    constexpr int QG = NTT_EPT / 3 * 3;
    if (tw_pass && NTT_TWPASS_HALVES) {
        // all of the thread's inter-pass twiddles are requested before the first product: two memory latencies per tile instead of
        // one per group of three (all sixteen at once made the compiler spill: it hoists the loads above the last stage)
        constexpr int HALF = NTT_EPT / 2, HG = HALF / 3 * 3;
#pragma unroll 1
        for (int h = 0; h < NTT_EPT; h += HALF) {
            gl_t x[HALF], tw[HALF]; uint32_t o[HALF];
#pragma unroll
            for (int q = 0; q < HALF; q++) elem(h + q, x[q], tw[q], o[q]);
#pragma unroll
            for (int q = 0; q < HG; q += 3) glx_mul3<true>(x[q], tw[q], x[q + 1], tw[q + 1], x[q + 2], tw[q + 2], x[q], x[q + 1], x[q + 2]);
#pragma unroll
            for (int q = HG; q < HALF; q++) x[q] = glx_mul<true>(x[q], tw[q]);
#pragma unroll
            for (int q = 0; q < HALF; q++) if (!NTT_DBG(p, 4) || x[q] == 12345) ntt_st(dst, o[q], x[q]);
        }
    } else {
#pragma unroll 1
    for (int q = 0; q < QG; q += 3) {
        gl_t x[3], tw[3]; uint32_t o[3];
#pragma unroll
        for (int k = 0; k < 3; k++) elem(q + k, x[k], tw[k], o[k]);
        glx_mul3<true>(x[0], tw[0], x[1], tw[1], x[2], tw[2], x[0], x[1], x[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) if (!NTT_DBG(p, 4) || x[k] == 12345) ntt_st(dst, o[k], x[k]);
    }
#pragma unroll
    for (int q = QG; q < NTT_EPT; q++) {
        gl_t x, tw; uint32_t o;
        elem(q, x, tw, o);
        x = glx_mul<true>(x, tw);
        if (!NTT_DBG(p, 4) || x == 12345) ntt_st(dst, o, x);
    }
    }
}
template <int LOGL, bool INV, bool ZP = false>
__global__ __launch_bounds__(NTT_THREADS, NTT_WAVES_PER_SIMD) void ntt_col_pass(NttPassParams p) {
    extern __shared__ __align__(16) gl_t lds[];
    ntt_col_tile<LOGL, INV, ZP>(p, lds, threadIdx.x, blockIdx.x, blockIdx.y);
}

// ROW pass (pass B, or the only pass when lgN1 == 0).
//   two-pass:   grid = (N1 / T, batch): rows k1 of one polynomial
//   single:     grid = (ceil(batch / T), 1): T polynomials
template <int LOGL, bool INV, bool SINGLE>
__device__ __forceinline__ void ntt_row_tile(const NttPassParams& p, gl_t* lds, const int tid, const uint32_t bx, const uint32_t by) {
    using G = NttGeom<LOGL>;
    constexpr int LOGT = G::LOGT, T = G::T, L = G::L;
    constexpr bool single = SINGLE;                         // == (p.lgN1 == 0)
    uint32_t tile = bx;
    if (!single && (gridDim.x & 7) == 0) tile = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const uint32_t out_shift = single ? 0 : p.out_shift;
    const bool nested = out_shift != 0;
    const uint32_t r0 = nested ? 0 : tile << LOGT;          // first row (k1) or first polynomial
    const uint32_t b = single ? 0 : (nested ? by << LOGT : by);        // (first) batch entry of the tile
    const gl_t* src = p.src + (uint64_t)b * p.src_stride;
    gl_t* dst = p.dst + (uint64_t)(b >> out_shift) * p.dst_stride;
    // row t of the tile starts at element row_off + (t << row_shift) of src, and its output k2 goes to k_off + t + (k2 << k_shift)
    const uint32_t row_shift = nested ? p.lgN1 + p.lgN2 : LOGL, row_off = nested ? tile << LOGL : r0 << LOGL;
    const uint32_t k_shift = p.lgN1 + out_shift, k_off = nested ? (tile << out_shift) + (b & ((1u << out_shift) - 1u)) : r0;
    if constexpr (G::WAVE_OWNED && !SINGLE && !NTT_ABLATION_BUILD) {
        // second pass over wave-owned rows: every wave reads its own rows (contiguous, canonical: the column pass wrote them)
        // straight into the registers of the first radix stage
        // The twiddle table is requested first and its barrier sits UNDER the latency of the row loads (memory returns in order: the
        // table words are there long before the rows): after it the waves of the workgroup run independently up to the final transpose.
        ntt_fill_lds_twiddles<LOGL>(lds, p.tw_local, tid);
        ntt_first_stage_direct<LOGL, INV, false, false>(lds, tid,
            [&](int t, int i2) -> gl_t { return ntt_ld(src, row_off + ((uint32_t)t << row_shift) + (uint32_t)i2); },
            [](int, int, gl_t v) -> gl_t { return v; },
            [] { __syncthreads(); });
        ntt_wave_sync();
        ntt_lds_stages<LOGL, ntt_first_radix(LOGL), INV>(lds, p.tw_local, tid);
    } else {
    // load: e -> (r = e / L, i2 = e % L): contiguous rows
    gl_t v[NTT_EPT];
#pragma unroll
    for (int q = 0; q < NTT_EPT; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t i2 = e & (L - 1), r = e >> LOGL;
        if (single) {
            const uint32_t poly = r0 + r;
            v[q] = (poly < p.batch && i2 < p.n_in) ? p.src[(uint64_t)poly * p.src_stride + i2] : 0;
        } else {
            v[q] = NTT_DBG(p, 2) ? (gl_t)e : ntt_ld(src, row_off + (r << row_shift) + i2);
        }
    }
#pragma unroll
    for (int q = 0; q < NTT_EPT; q++) {
        const int e = tid + NTT_THREADS * q;
        const uint32_t i2 = e & (L - 1), r = e >> LOGL;
        gl_t x = v[q];
        if (single) {       // two-pass: the column pass left canonical values
            if (p.pre_lo && i2 < p.n_in) x = glx_mul<true>(x, ntt_pow2level(p.pre_lo, p.pre_hi, i2));
            else x = glx_canon(x);
        }
        lds[G::at(r, i2)] = x;
    }
    ntt_fill_lds_twiddles<LOGL>(lds, p.tw_local, tid);
    __syncthreads();
    if (!NTT_DBG(p, 1)) ntt_lds_stages<LOGL, 0, INV>(lds, p.tw_local, tid);
    }
    if constexpr (G::WAVE_OWNED) __syncthreads();
    if (single) {
        // store rows contiguously: e -> (r = e / L, k = e % L)
#pragma unroll 4
        for (int q = 0; q < NTT_EPT; q++) {
            const int e = tid + NTT_THREADS * q;
            const uint32_t k = e & (L - 1), r = e >> LOGL;
            const uint32_t poly = r0 + r;
            if (poly < p.batch) {
                gl_t x = lds[G::at(r, k)];
                if (p.post_lo) x = glx_mul<true>(x, ntt_pow2level(p.post_lo, p.post_hi, k));
                else if (p.post_const != 1) x = glx_mul<true>(x, p.post_const);
                p.dst[(uint64_t)poly * p.dst_stride + k] = x;
            }
        }
    } else {
        // store X[k1 + N1*k2]: e -> (k2 = e / T, r = e % T): T-element segments
#pragma unroll 4
        for (int q = 0; q < NTT_EPT; q++) {
            const int e = tid + NTT_THREADS * q;
            const uint32_t r = e & (T - 1), k2 = e >> LOGT;
            const uint32_t k = k_off + r + (k2 << k_shift);
            gl_t x = lds[G::at(r, k2)];
            if (p.post_lo) x = glx_mul<true>(x, ntt_pow2level(p.post_lo, p.post_hi, k));
            else if (p.post_const != 1) x = glx_mul<true>(x, p.post_const);
            if (!NTT_DBG(p, 4) || x == 12345) ntt_st(dst, k, x);
        }
    }
}
template <int LOGL, bool INV, bool SINGLE>
__global__ __launch_bounds__(NTT_THREADS, NTT_WAVES_PER_SIMD) void ntt_row_pass(NttPassParams p) {
    extern __shared__ __align__(16) gl_t lds[];
    ntt_row_tile<LOGL, INV, SINGLE>(p, lds, threadIdx.x, blockIdx.x, blockIdx.y);
}

// out_lo[j] = base^j (j < 2^SPLIT), out_hi[j] = scale * base^(j << SPLIT) (j < hi_len)
__global__ void ntt_power_table(gl_t base, gl_t scale, gl_t* out_lo, gl_t* out_hi, uint32_t hi_len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo_len = 1u << NTT_SPLIT_LOG;
    if (j < lo_len) out_lo[j] = gl_canon(gl_exp(base, j));
    if (j < hi_len) out_hi[j] = gl_canon(gl_mul(scale, gl_exp(base, (uint64_t)j << NTT_SPLIT_LOG)));
}
// out[(k1 << lgN2) + i2] = scale * base^(i2 * k1): the inter-pass twiddles in the order the column pass stores its output
// (scale = the 1/N of an inverse transform, folded in here so that the row pass needs no multiply for it)
__global__ void ntt_pass_table(gl_t base, gl_t scale, gl_t* out, uint32_t lgN1, uint32_t lgN2) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (1u << (lgN1 + lgN2))) return;
    const uint32_t k1 = j >> lgN2, i2 = j & ((1u << lgN2) - 1);
    out[j] = gl_canon(gl_mul(scale, gl_exp(base, (uint64_t)i2 * k1)));
}
__global__ void ntt_root_table(gl_t base, gl_t* out, uint32_t len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < len) out[j] = gl_canon(gl_exp(base, j));
}
