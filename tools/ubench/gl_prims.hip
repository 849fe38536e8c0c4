// Cost of Goldilocks primitives on gfx950, candidate by candidate: each kernel runs ITERS x 8 independent dependent-chains per
// lane at 8 waves per SIMD and prints the VALU-pipe cycles one operation occupies (time x clock / operations per SIMD), next to
// the raw issue cost of the instructions the candidates are made of.  Every candidate is first checked against the host's
// unsigned __int128 arithmetic on random and edge inputs.
//   hipcc -O3 --offload-arch=gfx950 -I plonky2_demo_amd/csrc tools/ubench/gl_prims.hip -o /tmp/gl_prims && /tmp/gl_prims
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gl64.cuh"
#include "gl64_gfx950.cuh"

typedef uint32_t u32;
#define ITERS 512
#define NCH 8

// ---------------------------------------------------------------------------------------------------- candidates
__device__ __forceinline__ gl_t c_mul_compiler(gl_t a, gl_t b) { return gl_mul(a, b); }
__device__ __forceinline__ gl_t c_add_compiler(gl_t a, gl_t b) { return gl_add(a, b); }
__device__ __forceinline__ gl_t c_sub_compiler(gl_t a, gl_t b) { return gl_sub(a, b); }
__device__ __forceinline__ gl_t c_addc_compiler(gl_t a, gl_t b) { return gl_add_c(a, gl_canon(b)); }

__device__ __forceinline__ gl_t mk64(u32 lo, u32 hi) { return ((gl_t)hi << 32) | lo; }

// a + b, arbitrary representatives, carries in VCC (gfx950: 2 wait states between a VALU writing VCC and a VALU reading it)
__device__ __forceinline__ gl_t a_add(gl_t a, gl_t b) {
    u32 r0, r1, e;
    asm("v_add_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(e)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return mk64(r0, r1);
}
// a + b with b canonical: one correction
__device__ __forceinline__ gl_t a_add_c(gl_t a, gl_t b) {
    u32 r0, r1, e;
    asm("v_add_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(e)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return mk64(r0, r1);
}
// a - b, arbitrary representatives: borrow -> subtract EPS = (x0 - 0xFFFFFFFF, borrow into x1)
__device__ __forceinline__ gl_t a_sub(gl_t a, gl_t b) {
    u32 r0, r1, e;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(e)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    return mk64(r0, r1);
}

// ---- instruction-count-minimal forms (gfx950: every VALU instruction costs about the same ~4.2 cycles in a mixed stream, so the
// 64-bit instructions v_mad_u64_u32 / v_lshl_add_u64 / v_cmp_*_u64 are the cheap ones) ----------------------------------------
// a * b, any representatives in, any representative out: 4 mads (cross terms chained through the 64-bit addend, their carry in an
// SGPR pair), 3 carry adds, one mad for lo + hl * EPS, the subtraction of hh, and ONE two-sided fix-up: 14 VALU instructions.
template <bool CANON>
__device__ __forceinline__ gl_t m_mul_t(gl_t a, gl_t b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    gl_t p00, mid, p11;
    uint64_t cm;       // carry of the cross-term sum, worth 2^96 = -1
    asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1"
        : "=&v"(p00), "=&v"(mid), "=&v"(p11), "=&s"(cm)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    // product words: w0 = p00.lo, w1 = p00.hi + mid.lo, w2 = p11.lo + mid.hi + carry, w3 = p11.hi + carry (+ cm, kept apart).
    // w1 and w2 are computed in place (tied operands) so that (w0, w1) stays the register pair of p00: no moves
    u32 w1 = (u32)(p00 >> 32), w2 = (u32)p11, w3 = (u32)(p11 >> 32);
    asm("v_add_co_u32 %[w1], vcc, %[w1], %[m0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[w2], vcc, %[w2], %[m1], vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[w3], vcc"          // no carry out: the product is < 2^128
        : [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3)
        : [m0] "v"((u32)mid), [m1] "v"((u32)(mid >> 32))
        : "vcc");
    const gl_t lo = mk64((u32)p00, w1);
    gl_t z;
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(w2), "v"(lo));      // z = lo + hl * EPS, carry c
    // y = z - hh (hh = w3 + cm), borrow b.  True value = y + (c - b) 2^64: c > b: add EPS; b > c: subtract EPS (neither overflows)
    u32 y0 = (u32)z, y1 = (u32)(z >> 32), f0, f1;
    uint64_t t;
    if constexpr (!CANON) {
        asm("s_nop 1\n\t"
            "v_subb_co_u32_e64 %[y0], vcc, %[y0], %[w3], %[cm]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32 %[y1], vcc, 0, %[y1], vcc\n\t"
            "s_nop 1\n\t"
            "s_andn2_b64 %[t], vcc, %[c]\n\t"                           // N = b & ~c
            "s_andn2_b64 vcc, %[c], vcc\n\t"                            // P = c & ~b
            "v_cndmask_b32_e64 %[f1], 0, -1, %[t]\n\t"
            "v_cndmask_b32_e64 %[f0], 0, -1, vcc\n\t"
            "v_sub_u32 %[f0], %[f0], %[f1]"                             // (f1:f0) = +EPS, -EPS (mod 2^64) or 0
            : [y0] "+v"(y0), [y1] "+v"(y1), [f0] "=&v"(f0), [f1] "=&v"(f1), [t] "=&s"(t)
            : [w3] "v"(w3), [cm] "s"(cm), [c] "s"(c)
            : "vcc", "scc");
        return mk64(y0, y1) + mk64(f0, f1);
    } else {
        uint64_t bw;
        asm("s_nop 1\n\t"
            "v_subb_co_u32_e64 %[y0], vcc, %[y0], %[w3], %[cm]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32_e64 %[y1], %[bw], 0, %[y1], vcc"
            : [y0] "+v"(y0), [y1] "+v"(y1), [bw] "=&s"(bw) : [w3] "v"(w3), [cm] "s"(cm) : "vcc");
        // with f = +-EPS the result is already < p; with f = 0 it may not be: then EPS is added as well (y - p mod 2^64)
        const gl_t y = mk64(y0, y1);
        asm("s_nop 0\n\t"
            "v_cmp_gt_u64_e64 vcc, %[y], %[pm1]\n\t"
            "s_andn2_b64 %[t], %[bw], %[c]\n\t"                         // N = b & ~c
            "s_nop 0\n\t"
            "s_andn2_b64 vcc, vcc, %[t]\n\t"                            // (y >= p) & ~N
            "s_andn2_b64 %[bw], %[c], %[bw]\n\t"                        // P = c & ~b
            "s_or_b64 vcc, vcc, %[bw]\n\t"
            "v_cndmask_b32_e64 %[f1], 0, -1, %[t]\n\t"
            "v_cndmask_b32_e64 %[f0], 0, -1, vcc\n\t"
            "v_sub_u32 %[f0], %[f0], %[f1]"
            : [f0] "=&v"(f0), [f1] "=&v"(f1), [t] "=&s"(t), [bw] "+s"(bw)
            : [y] "v"(y), [pm1] "s"(GL_P - 1), [c] "s"(c)
            : "vcc", "scc");
        return y + mk64(f0, f1);
    }
}
__device__ __forceinline__ gl_t m_mul(gl_t a, gl_t b) { return m_mul_t<false>(a, b); }
__device__ __forceinline__ gl_t m_canon(gl_t x) { return x + ((x >= GL_P) ? GL_EPS : 0); }
__device__ __forceinline__ gl_t m_mul_c(gl_t a, gl_t b) { return m_mul_t<true>(a, b); }
// canonical + canonical -> canonical (5 VALU)
__device__ __forceinline__ gl_t m_add_cc(gl_t a, gl_t b) { const gl_t s = a + b; return s + (((s < a) | (s >= GL_P)) ? GL_EPS : 0); }
// canonical - canonical -> canonical (5 VALU): borrow -> subtract EPS (mod 2^64: + p)
__device__ __forceinline__ gl_t m_sub_cc(gl_t a, gl_t b) {
    u32 r0 = (u32)a, r1 = (u32)(a >> 32), e;
    asm("v_sub_co_u32 %[r0], vcc, %[r0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %[r1], vcc, %[r1], %[b1], vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[e]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "+v"(r0), [r1] "+v"(r1), [e] "=&v"(e)
        : [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc");
    return mk64(r0, r1);
}
// lo + top * EPS with one fix-up (top * EPS + lo < 2^64 + 2^64): v_mad_u64_u32 does the multiply, the 64-bit add and the carry
__device__ __forceinline__ gl_t m_reduce96(gl_t lo, u32 top) {
    gl_t z; uint64_t c; u32 e;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(top), "v"(lo));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(e) : "s"(c));
    return z + (gl_t)e;
}
__device__ __forceinline__ gl_t m_accred(gl_t al, gl_t ah) {
    u32 w1 = (u32)(al >> 32), top = (u32)(ah >> 32);
    asm("v_add_co_u32 %[w1], vcc, %[w1], %[ahl]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[top], vcc, 0, %[top], vcc"
        : [w1] "+v"(w1), [top] "+v"(top)
        : [ahl] "v"((u32)ah)
        : "vcc");
    return m_reduce96(mk64((u32)al, w1), top);
}
// z (+ carry c) -> canonical: add EPS (= subtract p mod 2^64) when the carry is set or z >= p
__device__ __forceinline__ gl_t m_fix_canon(gl_t z, uint64_t c) {
    u32 e; uint64_t ge;
    asm("s_nop 0\n\t"
        "v_cmp_gt_u64_e64 %[ge], %[z], %[pm1]\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %[ge], %[ge], %[c]\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, %[ge]"
        : [e] "=v"(e), [ge] "=&s"(ge) : [z] "v"(z), [pm1] "s"(GL_P - 1), [c] "s"(c) : "scc");
    return z + (gl_t)e;
}
// MDS accumulator pair -> field element (al, ah < 2^63): candidates for psd_acc_reduce
__device__ __forceinline__ gl_t c_accred_compiler(gl_t al, gl_t ah) {
    const u32 al_hi = (u32)(al >> 32), ah_lo = (u32)ah;
    const u32 mid = al_hi + ah_lo;
    const u32 top = (u32)(ah >> 32) + (mid < ah_lo ? 1u : 0u);
    return gl_reduce96(((gl_t)mid << 32) | (u32)al, top);
}
template <int E>
__device__ __forceinline__ gl_t c_shl_compiler(gl_t x) { return gl_mul_2exp(x, E); }

// ------------------------------------------------------------------------------------------------------ harness
template <typename F>
__global__ __launch_bounds__(256) void k_chain2(F f, const gl_t* in, gl_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gl_t x[NCH], y[NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) { x[k] = in[(i * NCH + k) % 4096]; y[k] = in[(i * NCH + k + 77) % 4096]; }
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < NCH; k++) x[k] = f(x[k], y[k]);
#pragma unroll
        for (int k = 0; k < NCH; k++) y[k] = f(y[k], x[k]);
    }
    gl_t s = 0;
#pragma unroll
    for (int k = 0; k < NCH; k++) s ^= x[k] ^ y[k];
    out[i] = s;
}
template <typename F>
__global__ __launch_bounds__(256) void k_chain1(F f, const gl_t* in, gl_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gl_t x[NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) x[k] = in[(i * NCH + k) % 4096];
#pragma unroll 1
    for (int it = 0; it < 2 * ITERS; it++) {
#pragma unroll
        for (int k = 0; k < NCH; k++) x[k] = f(x[k]);
    }
    gl_t s = 0;
#pragma unroll
    for (int k = 0; k < NCH; k++) s ^= x[k];
    out[i] = s;
}
// one application per element, for the correctness check
template <typename F>
__global__ void k_apply2(F f, const gl_t* a, const gl_t* b, gl_t* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = f(a[i], b[i]);
}
template <typename F>
__global__ void k_apply1(F f, const gl_t* a, gl_t* o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = f(a[i]);
}


template <bool CANON>
__global__ void k_apply_mul3(const gl_t* a, const gl_t* b, gl_t* o, size_t n) {
    const size_t i = 3 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 2 < n) glx_mul3<CANON>(a[i], b[i], a[i + 1], b[i + 1], a[i + 2], b[i + 2], o[i], o[i + 1], o[i + 2]);
}
template <bool CANON>
__global__ __launch_bounds__(256) void k_chain_mul3(const gl_t* in, gl_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gl_t x[9], y[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { x[k] = in[(i * 9 + k) % 4096]; y[k] = in[(i * 9 + k + 77) % 4096]; }
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 9; k += 3) glx_mul3<CANON>(x[k], y[k], x[k + 1], y[k + 1], x[k + 2], y[k + 2], x[k], x[k + 1], x[k + 2]);
#pragma unroll
        for (int k = 0; k < 9; k += 3) glx_mul3<CANON>(y[k], x[k], y[k + 1], x[k + 1], y[k + 2], x[k + 2], y[k], y[k + 1], y[k + 2]);
    }
    gl_t s = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s ^= x[k] ^ y[k];
    out[i] = s;
}
// the S-box layer of a full Poseidon round on 12 words: x^7 = x * x^2 * x^4 (plonky2/src/hash/poseidon.rs:240-261)
template <int V>
__global__ __launch_bounds__(256) void k_chain_sbox12(const gl_t* in, gl_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    gl_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = in[(i * 12 + k) % 4096];
#pragma unroll 1
    for (int it = 0; it < ITERS / 4; it++) {
        if constexpr (V == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) { gl_t x2 = gl_mul(s[k], s[k]), x4 = gl_mul(x2, x2), x3 = gl_mul(s[k], x2); s[k] = gl_mul(x3, x4); }
        } else if constexpr (V == 1) {
#pragma unroll
            for (int k = 0; k < 12; k++) { gl_t x2 = m_mul(s[k], s[k]), x4 = m_mul(x2, x2), x3 = m_mul(s[k], x2); s[k] = m_mul(x3, x4); }
        } else {
#pragma unroll
            for (int k = 0; k < 12; k += 3) {
                gl_t a2, b2, c2, a4, b4, c4, a3, b3, c3;
                glx_mul3<false>(s[k], s[k], s[k + 1], s[k + 1], s[k + 2], s[k + 2], a2, b2, c2);
                glx_mul3<false>(a2, a2, b2, b2, c2, c2, a4, b4, c4);
                glx_mul3<false>(s[k], a2, s[k + 1], b2, s[k + 2], c2, a3, b3, c3);
                glx_mul3<false>(a3, a4, b3, b4, c3, c4, s[k], s[k + 1], s[k + 2]);
            }
        }
    }
    gl_t r = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) r ^= s[k];
    out[i] = r;
}

#define FN2(name) struct F_##name { __device__ __forceinline__ gl_t operator()(gl_t a, gl_t b) const { return name(a, b); } }
#define FN1(name, ...) struct F1_##name { __device__ __forceinline__ gl_t operator()(gl_t a) const { return __VA_ARGS__(a); } }
FN2(c_mul_compiler); FN2(m_mul); FN2(m_mul_c); FN2(m_add_cc); FN2(m_sub_cc); FN2(m_accred); FN2(c_add_compiler); FN2(a_add); FN2(c_sub_compiler); FN2(a_sub);
FN2(c_addc_compiler); FN2(a_add_c); FN2(c_accred_compiler);
FN1(c12, c_shl_compiler<12>); FN1(c24, c_shl_compiler<24>); 
FN1(m12, glx_shl_c<12>); FN1(m24, glx_shl_c<24>); FN1(m32, glx_shl_c<32>); FN1(m36, glx_shl_c<36>); FN1(m48, glx_shl_c<48>); FN1(m60, glx_shl_c<60>); FN1(m72, glx_shl_c<72>); FN1(m84, glx_shl_c<84>); FN1(c72, c_shl_compiler<72>);
FN1(c36, c_shl_compiler<36>); FN1(c60, c_shl_compiler<60>); FN1(c84, c_shl_compiler<84>); FN1(c48, c_shl_compiler<48>);

static const uint64_t P = 0xFFFFFFFF00000001ULL;
static uint64_t h_mod(unsigned __int128 x) { return (uint64_t)(x % P); }
static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

static double g_clock_ghz = 2.4;
template <typename K, typename... A>
static double time_kernel(K k, int blocks, A... args) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::vector<uint64_t> ha, hb;
    const uint64_t edge[] = {0, 1, 2, P - 1, P, P + 1, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFEFFFFFFFFULL,
                             0x8000000000000000ULL, 0x7FFFFFFFFFFFFFFFULL, 0xFFFFFFFF, 0xFFFFFFFE00000001ULL, 0x00000001FFFFFFFFULL};
    // + the boundary forms of every glx_shl_c branch: 2^k for all k, h 2^(96-E) (x << E has a zero low word and a
    // zero middle word: the borrow-without-carry case), low (64 - E) bits zero, p - 2^k
    std::vector<uint64_t> grid(edge, edge + sizeof(edge) / sizeof(edge[0]));
    for (int k = 0; k < 64; k++) { grid.push_back(1ULL << k); grid.push_back(P - (1ULL << k)); }
    for (int E : {36, 48, 60}) for (uint64_t h : {1ULL, 3ULL, 7ULL, (1ULL << (E - 33)) + 1, (1ULL << (E - 32)) - 1}) grid.push_back(h << (96 - E));
    for (int E : {12, 24, 36, 48, 60}) grid.push_back((0xFEDCBA9876543210ULL >> (64 - E)) << (64 - E));
    for (uint64_t x : grid) for (uint64_t y : grid) { ha.push_back(x); hb.push_back(y); }
    uint64_t seed = 12345;
    while (ha.size() < (1u << 17)) { ha.push_back(splitmix(seed)); hb.push_back(splitmix(seed)); }
    const size_t n = ha.size();
    gl_t *da, *db, *dout, *din;
    hipMalloc((void**)&da, n * 8); hipMalloc((void**)&db, n * 8); hipMalloc((void**)&dout, (size_t)256 * 8 * 256 * 8 * 2); hipMalloc((void**)&din, 4096 * 8);
    hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(din, ha.data() + ha.size() - 4096, 4096 * 8, hipMemcpyHostToDevice);
    std::vector<uint64_t> ho(n);
    int bad_total = 0;
    std::vector<uint64_t> hac(ha), hbc(hb);
    for (auto& v : hac) v %= P;
    for (auto& v : hbc) v %= P;
    gl_t *dac, *dbc, *dinc;
    (void)hipMalloc((void**)&dac, n * 8); (void)hipMalloc((void**)&dbc, n * 8); (void)hipMalloc((void**)&dinc, 4096 * 8);
    (void)hipMemcpy(dac, hac.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dbc, hbc.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dinc, hac.data() + hac.size() - 4096, 4096 * 8, hipMemcpyHostToDevice);
    // canon_in: 0 = any representatives, 1 = b canonical, 2 = both canonical; canon_out: the result must be < p
    auto check2 = [&](const char* name, auto f, auto ref, int canon_in, bool canon_out) {
        const gl_t* pa = canon_in >= 2 ? dac : da; const gl_t* pb = canon_in >= 1 ? dbc : db;
        hipLaunchKernelGGL(k_apply2<decltype(f)>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f, pa, pb, dout, n);
        (void)hipMemcpy(ho.data(), dout, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t aa = canon_in >= 2 ? hac[i] : ha[i], bb = canon_in >= 1 ? hbc[i] : hb[i];
            const bool ok = (ho[i] % P == ref(aa, bb)) && (!canon_out || ho[i] < P);
            if (!ok) { if (bad < 3) printf("  %s MISMATCH a=%016llx b=%016llx got=%016llx want=%016llx\n", name, (unsigned long long)aa, (unsigned long long)bb, (unsigned long long)ho[i], (unsigned long long)ref(aa, bb)); bad++; }
        }
        bad_total += bad;
        return bad;
    };
    auto check1 = [&](const char* name, auto f, auto ref, bool canon_in, bool canon_out) {
        hipLaunchKernelGGL(k_apply1<decltype(f)>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f, canon_in ? dac : da, dout, n);
        (void)hipMemcpy(ho.data(), dout, n * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t aa = canon_in ? hac[i] : ha[i];
            if (ho[i] % P != ref(aa) || (canon_out && ho[i] >= P)) { if (bad < 3) printf("  %s MISMATCH a=%016llx got=%016llx want=%016llx\n", name, (unsigned long long)aa, (unsigned long long)ho[i], (unsigned long long)ref(aa)); bad++; }
        }
        bad_total += bad;
        return bad;
    };
    auto rmul = [](uint64_t a, uint64_t b) { return h_mod((unsigned __int128)a * b); };
    auto radd = [](uint64_t a, uint64_t b) { return h_mod((unsigned __int128)a + b); };
    auto rsub = [](uint64_t a, uint64_t b) { return h_mod((unsigned __int128)a + (unsigned __int128)P * 2 - b % P); };
    auto racc = [](uint64_t al, uint64_t ah) { return h_mod((unsigned __int128)(al >> 1) + ((unsigned __int128)(ah >> 1) << 32)); };
    const int blocks = 256 * 8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    const double ops2 = 2.0 * ITERS * NCH * 8;   // operations per SIMD (8 waves, each 2 * ITERS * NCH wave-wide operations)
    printf("%-28s %8s %12s %10s\n", "primitive", "check", "ns/op/SIMD", "cycles@2.4");
#define RUN2(name, ref, cin, cout)                                                                                    \
    {                                                                                                                 \
        int bad = check2(#name, F_##name(), ref, cin, cout);                                                          \
        double t = time_kernel(k_chain2<F_##name>, blocks, F_##name(), dinc, dout);                                   \
        printf("%-28s %8s %12.3f %10.1f\n", #name, bad ? "FAIL" : "ok", t / ops2 * 1e9, t / ops2 * g_clock_ghz * 1e9); \
    }
#define RUN1(name, ref, cin, cout)                                                                                    \
    {                                                                                                                 \
        int bad = check1(#name, F1_##name(), ref, cin, cout);                                                         \
        double t = time_kernel(k_chain1<F1_##name>, blocks, F1_##name(), dinc, dout);                                 \
        printf("%-28s %8s %12.3f %10.1f\n", #name, bad ? "FAIL" : "ok", t / ops2 * 1e9, t / ops2 * g_clock_ghz * 1e9); \
    }
    RUN2(c_mul_compiler, rmul, 0, false);
    RUN2(m_mul, rmul, 0, false);
    RUN2(m_mul_c, rmul, 0, true);
    RUN2(c_add_compiler, radd, 0, false);
    RUN2(a_add, radd, 0, false);
    RUN2(c_addc_compiler, radd, 1, false);
    RUN2(a_add_c, radd, 1, false);
    RUN2(m_add_cc, radd, 2, true);
    RUN2(c_sub_compiler, rsub, 0, false);
    RUN2(a_sub, rsub, 0, false);
    RUN2(m_sub_cc, rsub, 2, true);
    for (int canon = 0; canon < 2; canon++) {
        const size_t n3 = n / 3 * 3;
        if (canon) hipLaunchKernelGGL(k_apply_mul3<true>, dim3((unsigned)((n3 / 3 + 255) / 256)), dim3(256), 0, 0, da, db, dout, n3);
        else hipLaunchKernelGGL(k_apply_mul3<false>, dim3((unsigned)((n3 / 3 + 255) / 256)), dim3(256), 0, 0, da, db, dout, n3);
        (void)hipMemcpy(ho.data(), dout, n3 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < n3; i++) if (ho[i] % P != rmul(ha[i], hb[i]) || (canon && ho[i] >= P)) { if (bad < 3) printf("  mul3 MISMATCH a=%016llx b=%016llx got=%016llx\n", (unsigned long long)ha[i], (unsigned long long)hb[i], (unsigned long long)ho[i]); bad++; }
        bad_total += bad;
        double t = canon ? time_kernel(k_chain_mul3<true>, blocks, dinc, dout) : time_kernel(k_chain_mul3<false>, blocks, dinc, dout);
        const double ops3 = 2.0 * ITERS * 9 * 8;
        printf("%-28s %8s %12.3f %10.1f\n", canon ? "glx_mul3<canonical>" : "glx_mul3", bad ? "FAIL" : "ok", t / ops3 * 1e9, t / ops3 * g_clock_ghz * 1e9);
    }
    {
        const double sb = (ITERS / 4) * 12.0 * 8;      // S-boxes per SIMD
        double t0 = time_kernel(k_chain_sbox12<0>, blocks, dinc, dout), t1 = time_kernel(k_chain_sbox12<1>, blocks, dinc, dout), t2 = time_kernel(k_chain_sbox12<2>, blocks, dinc, dout);
        printf("%-28s %8s %12.3f %10.1f\n", "sbox x^7: gl_mul", "-", t0 / sb * 1e9, t0 / sb * g_clock_ghz * 1e9);
        printf("%-28s %8s %12.3f %10.1f\n", "sbox x^7: m_mul", "-", t1 / sb * 1e9, t1 / sb * g_clock_ghz * 1e9);
        printf("%-28s %8s %12.3f %10.1f\n", "sbox x^7: glx_mul3", "-", t2 / sb * 1e9, t2 / sb * g_clock_ghz * 1e9);
    }
    {
        struct G1 { __device__ __forceinline__ gl_t operator()(gl_t a, gl_t b) const { return c_accred_compiler(a >> 1, b >> 1); } };
        struct G2 { __device__ __forceinline__ gl_t operator()(gl_t a, gl_t b) const { return m_accred(a >> 1, b >> 1); } };
        int bad1 = check2("c_accred", G1(), racc, 0, false), bad2 = check2("m_accred", G2(), racc, 0, false);
        double t1 = time_kernel(k_chain2<G1>, blocks, G1(), din, dout), t2 = time_kernel(k_chain2<G2>, blocks, G2(), din, dout);
        printf("%-28s %8s %12.3f %10.1f   (includes two 64-bit shifts)\n", "c_accred_compiler", bad1 ? "FAIL" : "ok", t1 / ops2 * 1e9, t1 / ops2 * g_clock_ghz * 1e9);
        printf("%-28s %8s %12.3f %10.1f   (includes two 64-bit shifts)\n", "m_accred", bad2 ? "FAIL" : "ok", t2 / ops2 * 1e9, t2 / ops2 * g_clock_ghz * 1e9);
    }
    auto rsh = [](int e) { return [e](uint64_t a) { unsigned __int128 v = a % P; for (int i = 0; i < e; i++) v = (v * 2) % P; return (uint64_t)v; }; };
    RUN1(c12, rsh(12), false, false); RUN1(m12, rsh(12), true, true); RUN1(c24, rsh(24), false, false); RUN1(m24, rsh(24), true, true);
    RUN1(m32, rsh(32), true, true);
    RUN1(c36, rsh(36), false, false); RUN1(m36, rsh(36), true, true); RUN1(c48, rsh(48), false, false); RUN1(m48, rsh(48), true, true);
    RUN1(c60, rsh(60), false, false); RUN1(m60, rsh(60), true, true); RUN1(c72, rsh(72), false, false); RUN1(m72, rsh(72), true, true);
    RUN1(c84, rsh(84), false, false); RUN1(m84, rsh(84), true, true);
    printf("total mismatches: %d\n", bad_total);
    return bad_total ? 1 : 0;
}
