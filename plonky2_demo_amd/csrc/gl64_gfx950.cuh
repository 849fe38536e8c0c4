// Goldilocks arithmetic hand-scheduled for gfx950 (device only): products three at a time, canonical add / subtract / shift.
//
// What the instruction stream costs on this chip (tools/ubench/valu_class.hip, valu_mix.hip, gl_prims.hip; profiles/README.md):
// the 64-bit / multiply / compare instructions (v_mad_u64_u32, v_lshl_add_u64, v_cmp_*, v_lshlrev_b32 ...) take ~4.2 cycles of the
// SIMD each; the 32-bit add / sub / carry / select / logic instructions take ~2.4 -- but only in runs of their own kind -- and
// a VALU instruction that reads a carry or mask another VALU instruction wrote needs two instructions in between (else s_nop).
// One product is 4 multiply-adds for the 128-bit result, 3 carry adds, 1 multiply-add for lo + hl * EPS (the multiply by
// EPS = 2^32 - 1, the 64-bit add and its carry in one instruction), 2 borrow subtractions for - hh and one two-sided fix-up
// built from the two carry masks on the scalar unit: 15 instructions instead of the compiler's 24 -- and with THREE
// independent products interleaved the carry chains need no wait states and the cheap instructions form runs.
//
// Same function as gl_mul (field/src/goldilocks_field.rs:355-369 reduce128): any u64 representatives in, a u64
// representative out (CANON: the canonical one).
#pragma once
#include "gl64.cuh"

#if defined(__HIP_DEVICE_COMPILE__)
// (hi:lo) as a register pair.  Written as a two-lane vector, not (hi << 32) | lo: the compiler re-associates x + ((hi << 32) | lo)
// into (x + (hi << 32)) + lo -- two 64-bit adds and two zero registers instead of one add of the pair.
typedef uint32_t glx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gl_t glx_mk64(uint32_t lo, uint32_t hi) { glx_u32x2 v = {lo, hi}; return __builtin_bit_cast(gl_t, v); }
// z + bit * EPS (mod 2^64) for bit in {0, 1}: one multiply-add, no 64-bit pair to build for the addend
__device__ __forceinline__ gl_t glx_add_eps_if(gl_t z, uint32_t bit) {
    gl_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, -1, %2" : "=v"(r) : "v"(bit), "v"(z) : "vcc");
    return r;
}


template <bool CANON>
__device__ __forceinline__ void glx_mul3(gl_t aA, gl_t bA, gl_t aB, gl_t bB, gl_t aC, gl_t bC, gl_t& rA, gl_t& rB, gl_t& rC) {
    typedef uint32_t u32;
    gl_t p00A, midA, p11A, p00B, midB, p11B, p00C, midC, p11C;
    uint64_t cmA, cmB, cmC;              // carries of the cross-term sums, worth 2^96 = -1 each
    asm("v_mad_u64_u32 %[p00A], vcc, %[a0A], %[b0A], 0\n\t"
        "v_mad_u64_u32 %[p00B], vcc, %[a0B], %[b0B], 0\n\t"
        "v_mad_u64_u32 %[p00C], vcc, %[a0C], %[b0C], 0\n\t"
        "v_mad_u64_u32 %[midA], vcc, %[a0A], %[b1A], 0\n\t"
        "v_mad_u64_u32 %[midB], vcc, %[a0B], %[b1B], 0\n\t"
        "v_mad_u64_u32 %[midC], vcc, %[a0C], %[b1C], 0\n\t"
        "v_mad_u64_u32 %[p11A], vcc, %[a1A], %[b1A], 0\n\t"
        "v_mad_u64_u32 %[p11B], vcc, %[a1B], %[b1B], 0\n\t"
        "v_mad_u64_u32 %[p11C], vcc, %[a1C], %[b1C], 0\n\t"
        "v_mad_u64_u32 %[midA], %[cmA], %[a1A], %[b0A], %[midA]\n\t"
        "v_mad_u64_u32 %[midB], %[cmB], %[a1B], %[b0B], %[midB]\n\t"
        "v_mad_u64_u32 %[midC], %[cmC], %[a1C], %[b0C], %[midC]"
        : [p00A] "=&v"(p00A), [midA] "=&v"(midA), [p11A] "=&v"(p11A), [cmA] "=&s"(cmA),
          [p00B] "=&v"(p00B), [midB] "=&v"(midB), [p11B] "=&v"(p11B), [cmB] "=&s"(cmB),
          [p00C] "=&v"(p00C), [midC] "=&v"(midC), [p11C] "=&v"(p11C), [cmC] "=&s"(cmC)
        : [a0A] "v"((u32)aA), [a1A] "v"((u32)(aA >> 32)), [b0A] "v"((u32)bA), [b1A] "v"((u32)(bA >> 32)),
          [a0B] "v"((u32)aB), [a1B] "v"((u32)(aB >> 32)), [b0B] "v"((u32)bB), [b1B] "v"((u32)(bB >> 32)),
          [a0C] "v"((u32)aC), [a1C] "v"((u32)(aC >> 32)), [b0C] "v"((u32)bC), [b1C] "v"((u32)(bC >> 32))
        : "vcc");
    // product words: w0 = p00.lo, w1 = p00.hi + mid.lo, w2 = p11.lo + mid.hi + carry, w3 = p11.hi + carry (+ cm, kept apart).
    // w1 is computed in place (tied operand): (w0, w1) stays the register pair of p00 and feeds the next multiply-add without a move.
    // w2 and w3 are only ever used as single words, so they are FRESH outputs: tying both halves of p11's pair costs one register copy
    // per product (the allocator splits the pair and rejoins only one half).
    u32 w1A = (u32)(p00A >> 32), w1B = (u32)(p00B >> 32), w1C = (u32)(p00C >> 32), w2A, w3A, w2B, w3B, w2C, w3C;
    uint64_t sB, sC;
    asm("v_add_co_u32 %[w1A], vcc, %[w1A], %[m0A]\n\t"
        "v_add_co_u32_e64 %[w1B], %[sB], %[w1B], %[m0B]\n\t"
        "v_add_co_u32_e64 %[w1C], %[sC], %[w1C], %[m0C]\n\t"
        "v_addc_co_u32 %[w2A], vcc, %[q0A], %[m1A], vcc\n\t"
        "v_addc_co_u32_e64 %[w2B], %[sB], %[q0B], %[m1B], %[sB]\n\t"
        "v_addc_co_u32_e64 %[w2C], %[sC], %[q0C], %[m1C], %[sC]\n\t"
        "v_addc_co_u32 %[w3A], vcc, 0, %[q1A], vcc\n\t"                 // no carry out: a product is < 2^128
        "v_addc_co_u32_e64 %[w3B], %[sB], 0, %[q1B], %[sB]\n\t"
        "v_addc_co_u32_e64 %[w3C], %[sC], 0, %[q1C], %[sC]"
        : [w1A] "+v"(w1A), [w2A] "=&v"(w2A), [w3A] "=&v"(w3A), [w1B] "+v"(w1B), [w2B] "=&v"(w2B), [w3B] "=&v"(w3B),
          [w1C] "+v"(w1C), [w2C] "=&v"(w2C), [w3C] "=&v"(w3C), [sB] "=&s"(sB), [sC] "=&s"(sC)
        : [m0A] "v"((u32)midA), [m1A] "v"((u32)(midA >> 32)), [m0B] "v"((u32)midB), [m1B] "v"((u32)(midB >> 32)),
          [m0C] "v"((u32)midC), [m1C] "v"((u32)(midC >> 32)),
          [q0A] "v"((u32)p11A), [q1A] "v"((u32)(p11A >> 32)), [q0B] "v"((u32)p11B), [q1B] "v"((u32)(p11B >> 32)),
          [q0C] "v"((u32)p11C), [q1C] "v"((u32)(p11C >> 32))
        : "vcc");
    // z = lo + hl * EPS (mod 2^64), carry c
    const gl_t loA = glx_mk64((u32)p00A, w1A), loB = glx_mk64((u32)p00B, w1B), loC = glx_mk64((u32)p00C, w1C);
    gl_t zA, zB, zC;
    uint64_t cA, cB, cC;
    asm("v_mad_u64_u32 %[zA], %[cA], %[w2A], -1, %[loA]\n\t"
        "v_mad_u64_u32 %[zB], %[cB], %[w2B], -1, %[loB]\n\t"
        "v_mad_u64_u32 %[zC], %[cC], %[w2C], -1, %[loC]"
        : [zA] "=&v"(zA), [cA] "=&s"(cA), [zB] "=&v"(zB), [cB] "=&s"(cB), [zC] "=&v"(zC), [cC] "=&s"(cC)
        : [w2A] "v"(w2A), [loA] "v"(loA), [w2B] "v"(w2B), [loB] "v"(loB), [w2C] "v"(w2C), [loC] "v"(loC));
    // y = z - hh (hh = w3 + cm) mod 2^64, borrow b.  The true value is y + (c - b) 2^64: c > b -> add EPS, b > c -> subtract EPS
    // (neither overflows: goldilocks_field.rs:355-369); both results are < p, with c = b the value y may still be >= p.
    // The halves of y are fresh outputs (not tied to z's): tied halves of a 64-bit pair cost two register copies per product.
    u32 y0A, y1A, y0B, y1B, y0C, y1C;
    uint64_t bwA, bwB, bwC;
    asm("v_subb_co_u32_e64 %[y0A], vcc, %[z0A], %[w3A], %[cmA]\n\t"
        "v_subb_co_u32_e64 %[y0B], %[bwB], %[z0B], %[w3B], %[cmB]\n\t"
        "v_subb_co_u32_e64 %[y0C], %[bwC], %[z0C], %[w3C], %[cmC]\n\t"
        "v_subbrev_co_u32_e64 %[y1A], %[bwA], 0, %[z1A], vcc\n\t"
        "v_subbrev_co_u32_e64 %[y1B], %[bwB], 0, %[z1B], %[bwB]\n\t"
        "v_subbrev_co_u32_e64 %[y1C], %[bwC], 0, %[z1C], %[bwC]"
        : [y0A] "=&v"(y0A), [y1A] "=&v"(y1A), [y0B] "=&v"(y0B), [y1B] "=&v"(y1B), [y0C] "=&v"(y0C), [y1C] "=&v"(y1C),
          [bwA] "=&s"(bwA), [bwB] "=&s"(bwB), [bwC] "=&s"(bwC)
        : [z0A] "v"((u32)zA), [z1A] "v"((u32)(zA >> 32)), [z0B] "v"((u32)zB), [z1B] "v"((u32)(zB >> 32)), [z0C] "v"((u32)zC), [z1C] "v"((u32)(zC >> 32)),
          [w3A] "v"(w3A), [cmA] "s"(cmA), [w3B] "v"(w3B), [cmB] "s"(cmB), [w3C] "v"(w3C), [cmC] "s"(cmC)
        : "vcc");
    const gl_t yA = glx_mk64(y0A, y1A), yB = glx_mk64(y0B, y1B), yC = glx_mk64(y0C, y1C);
    u32 f0A, f1A, f0B, f1B, f0C, f1C;
    uint64_t tA, tB, tC;
    // N = b & ~c: subtract EPS.  P = c & ~b: add EPS.  CANON: y >= p and nothing subtracted: add EPS as well (= y - p mod 2^64).
    // (f1:f0) = +EPS, -EPS (mod 2^64) or 0; the scalar unit combines the masks while the vector unit is busy elsewhere.  The sum
    // y + f is one 64-bit add (v_lshl_add_u64), left to the compiler.
#define GLX_MUL3_TAIL                                                     \
        "v_cndmask_b32_e64 %[f1A], 0, -1, %[tA]\n\t"                      \
        "v_cndmask_b32_e64 %[f1B], 0, -1, %[tB]\n\t"                      \
        "v_cndmask_b32_e64 %[f1C], 0, -1, %[tC]\n\t"                      \
        "v_cndmask_b32_e64 %[f0A], 0, -1, %[bwA]\n\t"                     \
        "v_cndmask_b32_e64 %[f0B], 0, -1, %[bwB]\n\t"                     \
        "v_cndmask_b32_e64 %[f0C], 0, -1, %[bwC]\n\t"                     \
        "v_sub_u32 %[f0A], %[f0A], %[f1A]\n\t"                            \
        "v_sub_u32 %[f0B], %[f0B], %[f1B]\n\t"                            \
        "v_sub_u32 %[f0C], %[f0C], %[f1C]"
    if constexpr (CANON) {
        uint64_t gA, gB, gC;
        asm("v_cmp_gt_u64_e64 %[gA], %[yA], %[pm1]\n\t"
            "v_cmp_gt_u64_e64 %[gB], %[yB], %[pm1]\n\t"
            "v_cmp_gt_u64_e64 %[gC], %[yC], %[pm1]"
            : [gA] "=&s"(gA), [gB] "=&s"(gB), [gC] "=&s"(gC)
            : [yA] "v"(yA), [yB] "v"(yB), [yC] "v"(yC), [pm1] "s"(GL_P - 1));
        asm("s_andn2_b64 %[tA], %[bwA], %[cA]\n\t"
            "s_andn2_b64 %[tB], %[bwB], %[cB]\n\t"
            "s_andn2_b64 %[tC], %[bwC], %[cC]\n\t"
            "s_andn2_b64 %[bwA], %[cA], %[bwA]\n\t"
            "s_andn2_b64 %[bwB], %[cB], %[bwB]\n\t"
            "s_andn2_b64 %[bwC], %[cC], %[bwC]\n\t"
            "s_andn2_b64 %[gA], %[gA], %[tA]\n\t"
            "s_andn2_b64 %[gB], %[gB], %[tB]\n\t"
            "s_andn2_b64 %[gC], %[gC], %[tC]\n\t"
            "s_or_b64 %[bwA], %[gA], %[bwA]\n\t"
            "s_or_b64 %[bwB], %[gB], %[bwB]\n\t"
            "s_or_b64 %[bwC], %[gC], %[bwC]\n\t"
            GLX_MUL3_TAIL
            : [f0A] "=&v"(f0A), [f1A] "=&v"(f1A), [f0B] "=&v"(f0B), [f1B] "=&v"(f1B), [f0C] "=&v"(f0C), [f1C] "=&v"(f1C),
              [tA] "=&s"(tA), [tB] "=&s"(tB), [tC] "=&s"(tC), [gA] "+s"(gA), [gB] "+s"(gB), [gC] "+s"(gC),
              [bwA] "+s"(bwA), [bwB] "+s"(bwB), [bwC] "+s"(bwC)
            : [cA] "s"(cA), [cB] "s"(cB), [cC] "s"(cC)
            : "scc");
    } else {
        asm("s_andn2_b64 %[tA], %[bwA], %[cA]\n\t"
            "s_andn2_b64 %[tB], %[bwB], %[cB]\n\t"
            "s_andn2_b64 %[tC], %[bwC], %[cC]\n\t"
            "s_andn2_b64 %[bwA], %[cA], %[bwA]\n\t"
            "s_andn2_b64 %[bwB], %[cB], %[bwB]\n\t"
            "s_andn2_b64 %[bwC], %[cC], %[bwC]\n\t"
            GLX_MUL3_TAIL
            : [f0A] "=&v"(f0A), [f1A] "=&v"(f1A), [f0B] "=&v"(f0B), [f1B] "=&v"(f1B), [f0C] "=&v"(f0C), [f1C] "=&v"(f1C),
              [tA] "=&s"(tA), [tB] "=&s"(tB), [tC] "=&s"(tC), [bwA] "+s"(bwA), [bwB] "+s"(bwB), [bwC] "+s"(bwC)
            : [cA] "s"(cA), [cB] "s"(cB), [cC] "s"(cC)
            : "scc");
    }
#undef GLX_MUL3_TAIL
    rA = yA + glx_mk64(f0A, f1A); rB = yB + glx_mk64(f0B, f1B); rC = yC + glx_mk64(f0C, f1C);
}

// One product (the same arithmetic, with the wait states the lone carry chains need): for counts that are not multiples of three
template <bool CANON>
__device__ __forceinline__ gl_t glx_mul(gl_t a, gl_t b) {
    typedef uint32_t u32;
    gl_t p00, mid, p11;
    uint64_t cm;
    asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1"
        : "=&v"(p00), "=&v"(mid), "=&v"(p11), "=&s"(cm)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc");
    u32 w1 = (u32)(p00 >> 32), w2, w3;
    asm("v_add_co_u32 %[w1], vcc, %[w1], %[m0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[w2], vcc, %[q0], %[m1], vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[q1], vcc"
        : [w1] "+v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3)
        : [m0] "v"((u32)mid), [m1] "v"((u32)(mid >> 32)), [q0] "v"((u32)p11), [q1] "v"((u32)(p11 >> 32))
        : "vcc");
    const gl_t lo = glx_mk64((u32)p00, w1);
    gl_t z;
    uint64_t c, bw, t, g;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(w2), "v"(lo));
    u32 y0 = (u32)z, y1 = (u32)(z >> 32), f0, f1;
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[y0], vcc, %[y0], %[w3], %[cm]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e64 %[y1], %[bw], 0, %[y1], vcc"
        : [y0] "+v"(y0), [y1] "+v"(y1), [bw] "=&s"(bw) : [w3] "v"(w3), [cm] "s"(cm) : "vcc");
    if constexpr (CANON) {
        const gl_t y = glx_mk64(y0, y1);
        asm("v_cmp_gt_u64_e64 %0, %1, %2" : "=s"(g) : "v"(y), "s"(GL_P - 1));
        asm("s_nop 0\n\t"
            "s_andn2_b64 %[t], %[bw], %[c]\n\t"
            "s_andn2_b64 %[bw], %[c], %[bw]\n\t"
            "s_andn2_b64 %[g], %[g], %[t]\n\t"
            "s_or_b64 %[bw], %[g], %[bw]"
            : [t] "=&s"(t), [g] "+s"(g), [bw] "+s"(bw) : [c] "s"(c) : "scc");
    } else {
        asm("s_nop 0\n\t"
            "s_andn2_b64 %[t], %[bw], %[c]\n\t"
            "s_andn2_b64 %[bw], %[c], %[bw]"
            : [t] "=&s"(t), [bw] "+s"(bw) : [c] "s"(c) : "scc");
    }
    asm("v_cndmask_b32_e64 %[f1], 0, -1, %[t]\n\t"
        "v_cndmask_b32_e64 %[f0], 0, -1, %[bw]\n\t"
        "v_sub_u32 %[f0], %[f0], %[f1]\n\t"
        "v_add_co_u32 %[y0], vcc, %[y0], %[f0]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[y1], vcc, %[y1], %[f1], vcc"
        : [f0] "=&v"(f0), [f1] "=&v"(f1), [y0] "+v"(y0), [y1] "+v"(y1)
        : [t] "s"(t), [bw] "s"(bw)
        : "vcc");
    return glx_mk64(y0, y1);
}

// x * c + k for a compile-time c in [0, 64] and a wave-uniform 64-bit k: the first term of a multiply-add chain that starts
// from a constant.  The constant rides in as the scalar addend (c is an inline constant, so the instruction's one scalar
// operand is free for it); written in C the compiler first copies k into a vector register pair (two moves per chain).
__device__ __forceinline__ gl_t glx_mad_k(uint32_t x, uint32_t c /* a constant after inlining and unrolling */, uint64_t k) {
    gl_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "i"(c), "s"(k) : "vcc");
    return r;
}
// acc + x * c (mod 2^64) for a compile-time c in [0, 64], kept ONE multiply-add: written in C the compiler turns c = 2, 16 into a
// 64-bit shift-add, which first needs x zero-extended into a register pair (two moves per term)
__device__ __forceinline__ gl_t glx_mac_c(gl_t acc, uint32_t x, uint32_t c /* a constant after inlining and unrolling */) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "i"(c) : "vcc");
    return acc;
}
// lo + top * EPS (mod p) with one fix-up, valid while lo + top * EPS < 2^65 - 2^32 (goldilocks_field.rs:346-351 reduce96):
// v_mad_u64_u32 does the multiply by EPS, the 64-bit add and the carry in one instruction
__device__ __forceinline__ gl_t glx_reduce96(gl_t lo, uint32_t top) {
    gl_t z; uint64_t c; uint32_t e;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(top), "v"(lo));
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(e) : "s"(c));
    return glx_add_eps_if(z, e);
}
// al + ah 2^32 (al, ah < 2^63) -> one word: the Poseidon MDS accumulators
__device__ __forceinline__ gl_t glx_acc_reduce(gl_t al, gl_t ah) {
    uint32_t w1 = (uint32_t)(al >> 32), top = (uint32_t)(ah >> 32);
    asm("v_add_co_u32 %[w1], vcc, %[w1], %[ahl]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[top], vcc, 0, %[top], vcc"
        : [w1] "+v"(w1), [top] "+v"(top)
        : [ahl] "v"((uint32_t)ah)
        : "vcc");
    return glx_reduce96(glx_mk64((uint32_t)al, w1), top);
}
// three accumulator pairs at once (no wait states)
__device__ __forceinline__ void glx_acc_reduce3(gl_t alA, gl_t ahA, gl_t alB, gl_t ahB, gl_t alC, gl_t ahC, gl_t& rA, gl_t& rB, gl_t& rC) {
    typedef uint32_t u32;
    u32 w1A = (u32)(alA >> 32), topA = (u32)(ahA >> 32), w1B = (u32)(alB >> 32), topB = (u32)(ahB >> 32), w1C = (u32)(alC >> 32), topC = (u32)(ahC >> 32);
    uint64_t sB, sC;
    asm("v_add_co_u32 %[w1A], vcc, %[w1A], %[lA]\n\t"
        "v_add_co_u32_e64 %[w1B], %[sB], %[w1B], %[lB]\n\t"
        "v_add_co_u32_e64 %[w1C], %[sC], %[w1C], %[lC]\n\t"
        "v_addc_co_u32 %[tA], vcc, 0, %[tA], vcc\n\t"
        "v_addc_co_u32_e64 %[tB], %[sB], 0, %[tB], %[sB]\n\t"
        "v_addc_co_u32_e64 %[tC], %[sC], 0, %[tC], %[sC]"
        : [w1A] "+v"(w1A), [tA] "+v"(topA), [w1B] "+v"(w1B), [tB] "+v"(topB), [w1C] "+v"(w1C), [tC] "+v"(topC), [sB] "=&s"(sB), [sC] "=&s"(sC)
        : [lA] "v"((u32)ahA), [lB] "v"((u32)ahB), [lC] "v"((u32)ahC)
        : "vcc");
    const gl_t loA = glx_mk64((u32)alA, w1A), loB = glx_mk64((u32)alB, w1B), loC = glx_mk64((u32)alC, w1C);
    gl_t zA, zB, zC;
    uint64_t cA, cB, cC;
    u32 eA, eB, eC;
    asm("v_mad_u64_u32 %[zA], %[cA], %[tA], -1, %[loA]\n\t"
        "v_mad_u64_u32 %[zB], %[cB], %[tB], -1, %[loB]\n\t"
        "v_mad_u64_u32 %[zC], %[cC], %[tC], -1, %[loC]\n\t"
        "v_cndmask_b32_e64 %[eA], 0, 1, %[cA]\n\t"
        "v_cndmask_b32_e64 %[eB], 0, 1, %[cB]\n\t"
        "v_cndmask_b32_e64 %[eC], 0, 1, %[cC]"
        : [zA] "=&v"(zA), [cA] "=&s"(cA), [zB] "=&v"(zB), [cB] "=&s"(cB), [zC] "=&v"(zC), [cC] "=&s"(cC), [eA] "=&v"(eA), [eB] "=&v"(eB), [eC] "=&v"(eC)
        : [tA] "v"(topA), [loA] "v"(loA), [tB] "v"(topB), [loB] "v"(loB), [tC] "v"(topC), [loC] "v"(loC));
    rA = glx_add_eps_if(zA, eA); rB = glx_add_eps_if(zB, eB); rC = glx_add_eps_if(zC, eC);
}

// ---- canonical arithmetic for the NTT butterflies: operands < p in, results < p out (one fix-up each instead of two) ----------
__device__ __forceinline__ gl_t glx_canon(gl_t x) { return glx_add_eps_if(x, (x >= GL_P) ? 1u : 0u); }          // x - p mod 2^64
__device__ __forceinline__ gl_t glx_add_cc(gl_t a, gl_t b) {                                             // 5 instructions
    const gl_t s = a + b;
    return glx_add_eps_if(s, ((s < a) | (s >= GL_P)) ? 1u : 0u);
}
__device__ __forceinline__ gl_t glx_sub_cc(gl_t a, gl_t b) {                                             // 5 instructions
#ifdef GLX_C_BUTTERFLY
    const gl_t d = a - b;
    return d - ((a < b) ? GL_EPS : 0);
#endif
    // (outputs are NOT tied to a: a butterfly still needs a for a + b, and a tied operand would cost two register copies)
    uint32_t r0, r1, e;
    asm("v_sub_co_u32 %[r0], vcc, %[a0], %[b0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %[r1], vcc, %[a1], %[b1], vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"                        // borrow: + p = - EPS (mod 2^64)
        "v_sub_co_u32 %[r0], vcc, %[r0], %[e]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [e] "=&v"(e)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "vcc");
    return glx_mk64(r0, r1);
}
// four canonical differences at once: the borrow chains of four independent subtractions interleave, no wait states
__device__ __forceinline__ void glx_sub_cc4(const gl_t (&a)[4], const gl_t (&b)[4], gl_t (&r)[4]) {
    typedef uint32_t u32;
    u32 r0A, r1A, r0B, r1B, r0C, r1C, r0D, r1D, eA, eB, eC, eD;
    uint64_t sB, sC, sD;
    asm("v_sub_co_u32 %[r0A], vcc, %[a0A], %[b0A]\n\t"
        "v_sub_co_u32_e64 %[r0B], %[sB], %[a0B], %[b0B]\n\t"
        "v_sub_co_u32_e64 %[r0C], %[sC], %[a0C], %[b0C]\n\t"
        "v_sub_co_u32_e64 %[r0D], %[sD], %[a0D], %[b0D]\n\t"
        "v_subb_co_u32 %[r1A], vcc, %[a1A], %[b1A], vcc\n\t"
        "v_subb_co_u32_e64 %[r1B], %[sB], %[a1B], %[b1B], %[sB]\n\t"
        "v_subb_co_u32_e64 %[r1C], %[sC], %[a1C], %[b1C], %[sC]\n\t"
        "v_subb_co_u32_e64 %[r1D], %[sD], %[a1D], %[b1D], %[sD]\n\t"
        "v_cndmask_b32_e64 %[eA], 0, -1, vcc\n\t"                       // borrow: + p = - EPS (mod 2^64)
        "v_cndmask_b32_e64 %[eB], 0, -1, %[sB]\n\t"
        "v_cndmask_b32_e64 %[eC], 0, -1, %[sC]\n\t"
        "v_cndmask_b32_e64 %[eD], 0, -1, %[sD]\n\t"
        "v_sub_co_u32 %[r0A], vcc, %[r0A], %[eA]\n\t"
        "v_sub_co_u32_e64 %[r0B], %[sB], %[r0B], %[eB]\n\t"
        "v_sub_co_u32_e64 %[r0C], %[sC], %[r0C], %[eC]\n\t"
        "v_sub_co_u32_e64 %[r0D], %[sD], %[r0D], %[eD]\n\t"
        "v_subbrev_co_u32 %[r1A], vcc, 0, %[r1A], vcc\n\t"
        "v_subbrev_co_u32_e64 %[r1B], %[sB], 0, %[r1B], %[sB]\n\t"
        "v_subbrev_co_u32_e64 %[r1C], %[sC], 0, %[r1C], %[sC]\n\t"
        "v_subbrev_co_u32_e64 %[r1D], %[sD], 0, %[r1D], %[sD]"
        : [r0A] "=&v"(r0A), [r1A] "=&v"(r1A), [r0B] "=&v"(r0B), [r1B] "=&v"(r1B), [r0C] "=&v"(r0C), [r1C] "=&v"(r1C), [r0D] "=&v"(r0D), [r1D] "=&v"(r1D),
          [eA] "=&v"(eA), [eB] "=&v"(eB), [eC] "=&v"(eC), [eD] "=&v"(eD), [sB] "=&s"(sB), [sC] "=&s"(sC), [sD] "=&s"(sD)
        : [a0A] "v"((u32)a[0]), [a1A] "v"((u32)(a[0] >> 32)), [a0B] "v"((u32)a[1]), [a1B] "v"((u32)(a[1] >> 32)),
          [a0C] "v"((u32)a[2]), [a1C] "v"((u32)(a[2] >> 32)), [a0D] "v"((u32)a[3]), [a1D] "v"((u32)(a[3] >> 32)),
          [b0A] "v"((u32)b[0]), [b1A] "v"((u32)(b[0] >> 32)), [b0B] "v"((u32)b[1]), [b1B] "v"((u32)(b[1] >> 32)),
          [b0C] "v"((u32)b[2]), [b1C] "v"((u32)(b[2] >> 32)), [b0D] "v"((u32)b[3]), [b1D] "v"((u32)(b[3] >> 32))
        : "vcc");
    r[0] = glx_mk64(r0A, r1A); r[1] = glx_mk64(r0B, r1B); r[2] = glx_mk64(r0C, r1C); r[3] = glx_mk64(r0D, r1D);
}
// z (+ carry mask c) -> canonical: add EPS (= subtract p mod 2^64) when the carry is set or z >= p
__device__ __forceinline__ gl_t glx_fix_canon(gl_t z, uint64_t c) {
    uint32_t e; uint64_t ge;
    asm("s_nop 0\n\t"
        "v_cmp_gt_u64_e64 %[ge], %[z], %[pm1]\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %[ge], %[ge], %[c]\n\t"
        "v_cndmask_b32_e64 %[e], 0, 1, %[ge]"
        : [e] "=v"(e), [ge] "=&s"(ge) : [z] "v"(z), [pm1] "s"(GL_P - 1), [c] "s"(c) : "scc");
    return glx_add_eps_if(z, e);
}
// canonical x times 2^E (0 < E < 96), canonical result: the butterfly twiddles (the reference's w_64 = 2^39, 2^96 = -1)
template <int E>
__device__ __forceinline__ gl_t glx_shl_c(gl_t x) {
    static_assert(E > 0 && E < 96, "shift twiddle exponent");
#ifdef GLX_C_BUTTERFLY
    if constexpr (E < 32) {
        const gl_t ml = x << E;
        const gl_t t = (gl_t)((uint32_t)(x >> 32) >> (32 - E)) * 0xFFFFFFFFu;
        const gl_t zz = ml + t;
        return zz + (((zz < t) | (zz >= GL_P)) ? GL_EPS : 0);
    } else if constexpr (E >= 64) {
        constexpr int R = E - 64;
        const gl_t le = (gl_t)((uint32_t)x << R) * 0xFFFFFFFFu, hm = x >> (32 - R);
        const gl_t d = le - hm;
        return d - ((le < hm) ? GL_EPS : 0);
    }
#endif
    gl_t z; uint64_t c;
    if constexpr (E < 32) {                     // (h:m:l) = x << E: (m:l) + h EPS < 2^64 + 2^63
        const gl_t ml = x << E;
        const uint32_t h = (uint32_t)(x >> 32) >> (32 - E);
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"(h), "v"(ml));
        return glx_fix_canon(z, c);
    } else if constexpr (E == 32) {             // x0 2^32 + x1 2^64 = (x0 << 32) + x1 EPS, both canonical
        const gl_t a = x << 32;
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((uint32_t)(x >> 32)), "v"(a));
        return glx_fix_canon(z, c);
    } else if constexpr (E < 64) {              // x 2^E = a + m EPS - h: a = low 64 bits of x << E (= l 2^32 <= p - 1), (h:m) = x >> (64 - E)
        const gl_t a = x << E, u = x >> (64 - E);
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(c) : "v"((uint32_t)u), "v"(a));
        // y = z - h mod 2^64, borrow b; true value y + (c - b) 2^64 in (-2^31, 2^65): c > b: + EPS, b > c: - EPS, both then < p; else y may be >= p
        uint32_t y0 = (uint32_t)z, y1 = (uint32_t)(z >> 32), f0, f1;
        uint64_t bw, t, g;
        asm("s_nop 0\n\t"
            "v_sub_co_u32 %[y0], vcc, %[y0], %[h]\n\t"
            "s_nop 1\n\t"
            "v_subbrev_co_u32_e64 %[y1], %[bw], 0, %[y1], vcc"
            : [y0] "+v"(y0), [y1] "+v"(y1), [bw] "=&s"(bw) : [h] "v"((uint32_t)(u >> 32)) : "vcc");
        const gl_t y = glx_mk64(y0, y1);
        asm("v_cmp_gt_u64_e64 %[g], %[y], %[pm1]\n\t"
            "s_andn2_b64 %[t], %[bw], %[c]\n\t"
            "s_andn2_b64 %[bw], %[c], %[bw]\n\t"
            "s_nop 0\n\t"
            "s_andn2_b64 %[g], %[g], %[t]\n\t"
            "s_or_b64 %[bw], %[g], %[bw]\n\t"
            "v_cndmask_b32_e64 %[f1], 0, -1, %[t]\n\t"
            "v_cndmask_b32_e64 %[f0], 0, -1, %[bw]\n\t"
            "v_sub_u32 %[f0], %[f0], %[f1]"
            : [f0] "=&v"(f0), [f1] "=&v"(f1), [t] "=&s"(t), [g] "=&s"(g), [bw] "+s"(bw)
            : [y] "v"(y), [pm1] "s"(GL_P - 1), [c] "s"(c)
            : "scc");
        return y + glx_mk64(f0, f1);
    } else {                                    // (h:m:l) = x << (E - 64): l EPS - (h:m), both canonical
        constexpr int R = E - 64;
        const uint32_t l = (uint32_t)x << R;
        const gl_t hm = x >> (32 - R);
        gl_t le;
        asm("v_mad_u64_u32 %0, vcc, %1, -1, 0" : "=v"(le) : "v"(l) : "vcc");
        return glx_sub_cc(le, hm);
    }
}
// ---- sums of products without a reduction per term: the quotient's alpha-weighted constraint sums --------------------------
// sum_t term_t * alpha_t for TWO weight sequences (the two challenges alpha) kept as three 64-bit partial sums of 32 x 32-bit
// products each (weights 1, 2^32, 2^64) plus the number of times each wrapped: 16 instructions per term for both sums -- a
// modular multiply-add each would be 2 x ~21 -- and one reduction at the end.  Terms may be any u64 representatives; the
// weights must be wave-uniform (they are read as scalar operands).
struct GlxWideAcc2 {
    gl_t a0[2], a1[2], a2[2];       // sum t0*w0 | sum (t0*w1 + t1*w0) | sum t1*w1       (mod 2^64)
    uint32_t k0[2], k1[2], k2[2];   // wrap counts of the three sums (each worth 2^64 of its weight)
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int b = 0; b < 2; b++) { a0[b] = 0; a1[b] = 0; a2[b] = 0; k0[b] = 0; k1[b] = 0; k2[b] = 0; }
    }
    __device__ __forceinline__ void mac(gl_t term, gl_t wA, gl_t wB) {
        typedef uint32_t u32;
        uint64_t c0, c1, c2, c3, c4, c5;
        // the carry of every multiply-add is consumed at least three instructions later: no wait states
        asm("v_mad_u64_u32 %[a0A], %[c0], %[t0], %[wA0], %[a0A]\n\t"
            "v_mad_u64_u32 %[a0B], %[c1], %[t0], %[wB0], %[a0B]\n\t"
            "v_mad_u64_u32 %[a1A], %[c2], %[t0], %[wA1], %[a1A]\n\t"
            "v_mad_u64_u32 %[a1B], %[c3], %[t0], %[wB1], %[a1B]\n\t"
            "v_mad_u64_u32 %[a2A], %[c4], %[t1], %[wA1], %[a2A]\n\t"
            "v_mad_u64_u32 %[a2B], %[c5], %[t1], %[wB1], %[a2B]\n\t"
            "v_addc_co_u32_e64 %[k0A], vcc, 0, %[k0A], %[c0]\n\t"
            "v_addc_co_u32_e64 %[k0B], vcc, 0, %[k0B], %[c1]\n\t"
            "v_addc_co_u32_e64 %[k1A], vcc, 0, %[k1A], %[c2]\n\t"
            "v_addc_co_u32_e64 %[k1B], vcc, 0, %[k1B], %[c3]\n\t"
            "v_mad_u64_u32 %[a1A], %[c0], %[t1], %[wA0], %[a1A]\n\t"
            "v_mad_u64_u32 %[a1B], %[c1], %[t1], %[wB0], %[a1B]\n\t"
            "v_addc_co_u32_e64 %[k2A], vcc, 0, %[k2A], %[c4]\n\t"
            "v_addc_co_u32_e64 %[k2B], vcc, 0, %[k2B], %[c5]\n\t"
            "v_addc_co_u32_e64 %[k1A], vcc, 0, %[k1A], %[c0]\n\t"
            "v_addc_co_u32_e64 %[k1B], vcc, 0, %[k1B], %[c1]"
            : [a0A] "+v"(a0[0]), [a1A] "+v"(a1[0]), [a2A] "+v"(a2[0]), [k0A] "+v"(k0[0]), [k1A] "+v"(k1[0]), [k2A] "+v"(k2[0]),
              [a0B] "+v"(a0[1]), [a1B] "+v"(a1[1]), [a2B] "+v"(a2[1]), [k0B] "+v"(k0[1]), [k1B] "+v"(k1[1]), [k2B] "+v"(k2[1]),
              [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4), [c5] "=&s"(c5)
            : [t0] "v"((u32)term), [t1] "v"((u32)(term >> 32)),
              [wA0] "s"((u32)wA), [wA1] "s"((u32)(wA >> 32)), [wB0] "s"((u32)wB), [wB1] "s"((u32)(wB >> 32))
            : "vcc");
    }
    // canonical value of sum b:  a0 + a1 2^32 + (a2 + k0) 2^64 + k1 2^96 + k2 2^128,  2^64 = EPS, 2^96 = -1, 2^128 = -2^32 (mod p)
    __device__ __forceinline__ gl_t sum(int b) const {
        gl_t r = glx_add_cc(glx_canon(a0[b]), glx_shl_c<32>(glx_canon(a1[b])));
        r = glx_add_cc(r, glx_shl_c<64>(glx_canon(a2[b])));
        r = glx_add_cc(r, glx_shl_c<64>((gl_t)k0[b]));
        r = glx_sub_cc(r, (gl_t)k1[b]);
        return glx_sub_cc(r, glx_shl_c<32>((gl_t)k2[b]));
    }
};
#elif defined(__HIPCC__)
// host pass of a .hip file: kernels that call these must still parse
template <bool CANON>
__device__ void glx_mul3(gl_t aA, gl_t bA, gl_t aB, gl_t bB, gl_t aC, gl_t bC, gl_t& rA, gl_t& rB, gl_t& rC);
template <bool CANON>
__device__ gl_t glx_mul(gl_t a, gl_t b);
__device__ gl_t glx_reduce96(gl_t lo, uint32_t top);
__device__ gl_t glx_mad_k(uint32_t x, uint32_t c, uint64_t k);
__device__ gl_t glx_mac_c(gl_t acc, uint32_t x, uint32_t c);
__device__ gl_t glx_canon(gl_t x);
__device__ gl_t glx_add_eps_if(gl_t z, uint32_t bit);
__device__ gl_t glx_add_cc(gl_t a, gl_t b);
__device__ gl_t glx_sub_cc(gl_t a, gl_t b);
__device__ void glx_sub_cc4(const gl_t (&a)[4], const gl_t (&b)[4], gl_t (&r)[4]);
template <int E>
__device__ gl_t glx_shl_c(gl_t x);
__device__ gl_t glx_acc_reduce(gl_t al, gl_t ah);
__device__ void glx_acc_reduce3(gl_t alA, gl_t ahA, gl_t alB, gl_t ahB, gl_t alC, gl_t ahC, gl_t& rA, gl_t& rB, gl_t& rC);
struct GlxWideAcc2 {
    gl_t a0[2], a1[2], a2[2];
    uint32_t k0[2], k1[2], k2[2];
    __device__ void clear();
    __device__ void mac(gl_t term, gl_t wA, gl_t wB);
    __device__ gl_t sum(int b) const;
};
#endif
