// Goldilocks field (p = 2^64 - 2^32 + 1) for gfx950 kernels and for the host-side driver.
//
// gfx950 has no 64x64->128 multiplier: a product is four v_mad_u64_u32 (32x32+64) and the reduction
// uses 2^64 = 2^32 - 1, 2^96 = -1 (mod p).  Values are kept "weakly reduced": any u64 is a valid
// representative, exactly like the reference's GoldilocksField(pub u64)
// (field/src/goldilocks_field.rs:23-25); `gl_canon` gives the canonical one and every buffer that
// leaves the library is canonical.  Semantics mirror field/src/goldilocks_field.rs:199-274,346-369.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define GL_HD __host__ __device__ __forceinline__
#else
// host pass: let the x86 inliner decide (forcing everything inline makes the host Poseidon 1.7x slower)
#define GL_HD __host__ __device__ inline
#endif
#else
#define GL_HD inline
#endif

typedef uint64_t gl_t;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

GL_HD gl_t gl_canon(gl_t x) { return x >= GL_P ? x - GL_P : x; }

// a + b; correct for arbitrary u64 representatives
GL_HD gl_t gl_add(gl_t a, gl_t b) {
    gl_t s = a + b;
    gl_t c = (s < a) ? GL_EPS : 0;
    gl_t s2 = s + c;
    // second carry only possible when both inputs were non-canonical
    return s2 + ((s2 < c) ? GL_EPS : 0);
}
// a + b where b is canonical (b < p): one correction is enough
GL_HD gl_t gl_add_c(gl_t a, gl_t b) {
    gl_t s = a + b;
    return s + ((s < a) ? GL_EPS : 0);
}
GL_HD gl_t gl_sub(gl_t a, gl_t b) {
    gl_t d = a - b;
    gl_t c = (a < b) ? GL_EPS : 0;
    gl_t d2 = d - c;
    return d2 - ((d < c) ? GL_EPS : 0);
}
// a - b where b is canonical
GL_HD gl_t gl_sub_c(gl_t a, gl_t b) {
    gl_t d = a - b;
    return d - ((a < b) ? GL_EPS : 0);
}
GL_HD gl_t gl_neg(gl_t a) {
    gl_t c = gl_canon(a);
    return c ? GL_P - c : 0;
}
GL_HD gl_t gl_double(gl_t a) { return gl_add(a, a); }

// (lo + hi*2^64) mod p, hi < 2^32   (goldilocks_field.rs:346-351)
GL_HD gl_t gl_reduce96(gl_t lo, uint32_t hi) {
    gl_t t = ((gl_t)hi << 32) - hi;   // hi * (2^32 - 1) < 2^64
    gl_t s = lo + t;
    return s + ((s < t) ? GL_EPS : 0);
}
// (lo + hi*2^64) mod p   (goldilocks_field.rs:355-369)
GL_HD gl_t gl_reduce128(gl_t lo, gl_t hi) {
    uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
    gl_t t0 = lo - hh;
    t0 -= (lo < (gl_t)hh) ? GL_EPS : 0;
    gl_t t1 = ((gl_t)hl << 32) - hl;
    gl_t s = t0 + t1;
    return s + ((s < t1) ? GL_EPS : 0);
}

GL_HD void gl_mul_wide(gl_t a, gl_t b, gl_t& lo, gl_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    gl_t p00 = (gl_t)a0 * b0;
    gl_t m1 = (gl_t)a0 * b1 + (p00 >> 32);            // < 2^64
    gl_t m2 = (gl_t)a1 * b0 + (uint32_t)m1;           // < 2^64
    hi = (gl_t)a1 * b1 + (m1 >> 32) + (m2 >> 32);
    lo = (m2 << 32) | (uint32_t)p00;
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (gl_t)p; hi = (gl_t)(p >> 64);
#endif
}
GL_HD gl_t gl_mul(gl_t a, gl_t b) {
    gl_t lo, hi;
    gl_mul_wide(a, b, lo, hi);
    return gl_reduce128(lo, hi);
}
GL_HD gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
// acc + x*y  (goldilocks_field.rs:138-142)
GL_HD gl_t gl_mul_add(gl_t acc, gl_t x, gl_t y) {
    gl_t lo, hi;
    gl_mul_wide(x, y, lo, hi);
    gl_t l2 = lo + acc;
    hi += (l2 < lo) ? 1 : 0;     // cannot overflow: x*y + acc < 2^128
    return gl_reduce128(l2, hi);
}
// x * c for a small constant c < 2^32
GL_HD gl_t gl_mul_small(gl_t x, uint32_t c) {
    gl_t l = (gl_t)(uint32_t)x * c;
    gl_t h = (gl_t)(uint32_t)(x >> 32) * c + (l >> 32);   // < 2^64
    gl_t lo = (h << 32) | (uint32_t)l;
    return gl_reduce96(lo, (uint32_t)(h >> 32));
}

// x * 2^e mod p for 0 <= e < 192 (2 has order 192; 2^96 = -1).  With e a compile-time constant after
// unrolling this is a handful of shifts/adds: all radix-<=64 butterfly twiddles are of this form because
// the reference's 64th root of unity g^((p-1)/64) equals 2^39.
GL_HD gl_t gl_mul_2exp(gl_t x, unsigned e) {
    if (e >= 96) { x = gl_neg(x); e -= 96; }
    if (e == 0) return x;
    if (e < 32) return gl_reduce96(x << e, (uint32_t)(x >> (64 - e)));
    if (e == 32) return gl_reduce128(x << 32, x >> 32);
    if (e < 64) return gl_reduce128(x << e, x >> (64 - e));
    // e in [64, 96): x*2^e = (x*2^(e-64)) * 2^64
    gl_t y = (e == 64) ? x : gl_reduce96(x << (e - 64), (uint32_t)(x >> (128 - e)));
    // y * 2^64 = y * (2^32 - 1) = (y << 32) - y as a 96-bit number
    gl_t lo = y << 32, hi = y >> 32;
    gl_t l2 = lo - y;
    hi -= (lo < y) ? 1 : 0;
    return gl_reduce128(l2, hi);
}

GL_HD gl_t gl_exp(gl_t base, uint64_t e) {
    gl_t cur = base, prod = 1;
    while (e) {
        if (e & 1) prod = gl_mul(prod, cur);
        cur = gl_sqr(cur);
        e >>= 1;
    }
    return prod;
}
GL_HD gl_t gl_inv(gl_t a) { return gl_exp(a, GL_P - 2); }

// ---- quadratic extension F[X]/(X^2 - 7)  (field/src/extension/quadratic.rs:143-193) --------------
struct gl2_t {
    gl_t a, b;
};
GL_HD gl2_t gl2_make(gl_t a, gl_t b) { gl2_t r; r.a = a; r.b = b; return r; }
GL_HD gl2_t gl2_add(gl2_t x, gl2_t y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GL_HD gl2_t gl2_sub(gl2_t x, gl2_t y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GL_HD gl2_t gl2_mul(gl2_t x, gl2_t y) {
    gl_t bb = gl_mul(x.b, y.b);
    gl_t c0 = gl_add(gl_mul(x.a, y.a), gl_mul_small(bb, 7));
    gl_t c1 = gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a));
    return gl2_make(c0, c1);
}
GL_HD gl2_t gl2_scalar(gl2_t x, gl_t s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GL_HD gl2_t gl2_canon(gl2_t x) { return gl2_make(gl_canon(x.a), gl_canon(x.b)); }
GL_HD gl2_t gl2_inv(gl2_t x) {
    gl_t norm = gl_sub(gl_sqr(x.a), gl_mul_small(gl_sqr(x.b), 7));
    gl_t ni = gl_inv(norm);
    return gl2_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}
GL_HD gl2_t gl2_exp(gl2_t base, uint64_t e) {
    gl2_t cur = base, prod = gl2_make(1, 0);
    while (e) {
        if (e & 1) prod = gl2_mul(prod, cur);
        cur = gl2_mul(cur, cur);
        e >>= 1;
    }
    return prod;
}
