// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds), device + host.
// Same function as plonky2/src/hash/poseidon.rs:598-609 (pinned by the four known-answer vectors of
// plonky2/src/hash/poseidon_goldilocks.rs:449-485); the sponge / compression wrappers follow
// plonky2/src/hash/hashing.rs:98-146 and plonky2/src/plonk/config.rs:55-66.
//
// One lane owns one 12-word state (24 VGPRs).  The MDS layer exploits the 6-bit circulant entries:
// every state word is split into 32-bit halves and the two 12-term dot products are accumulated in
// 64-bit registers without intermediate reduction (v_mad_u64_u32 with SGPR constants), followed by
// one 96-bit reduction per output word -- the same lazy-reduction idea as poseidon.rs:176-198.
#pragma once
#include "gl64.cuh"
#include "poseidon_constants.h"

#if defined(__HIPCC__)
// device copies of the tables (per translation unit, constant address space -> scalar loads)
#define POSEIDON_TABLE(name, n) static __constant__ const uint64_t d_##name[n]
#include "poseidon_constants.inc"
#undef POSEIDON_TABLE
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PSD_TAB(name) d_##name
#else
#define PSD_TAB(name) name
#endif

GL_HD gl_t psd_sbox(gl_t x) {
    gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

GL_HD void psd_mds(gl_t (&s)[12]) {
    const uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
    gl_t out[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        gl_t al = 0, ah = 0;   // each < 12 * 41 * 2^32 < 2^41
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (gl_t)lo[(i + r) % 12] * circ[i];
            ah += (gl_t)hi[(i + r) % 12] * circ[i];
        }
        if (r == 0) { al += (gl_t)lo[0] * 8; ah += (gl_t)hi[0] * 8; }   // MDS_MATRIX_DIAG[0] = 8
        // value = al + ah * 2^32  (< 2^74)
        gl_t l = al + (ah << 32);
        uint32_t top = (uint32_t)(ah >> 32) + ((l < al) ? 1u : 0u);
        out[r] = gl_reduce96(l, top);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = out[i];
}

GL_HD void psd_full_round(gl_t (&s)[12], int round) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = psd_sbox(gl_add_c(s[i], PSD_TAB(POSEIDON_RC)[12 * round + i]));
    psd_mds(s);
}

GL_HD void psd_partial_rounds(gl_t (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], PSD_TAB(POSEIDON_PARTIAL_FIRST_RC)[i]);
    {
        gl_t t[12];
        t[0] = s[0];
#pragma unroll
        for (int c = 1; c < 12; c++) {
            gl_t acc = 0;
#pragma unroll
            for (int r = 1; r < 12; r++) acc = gl_mul_add(acc, s[r], PSD_TAB(POSEIDON_PARTIAL_INIT)[(r - 1) * 11 + (c - 1)]);
            t[c] = acc;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        gl_t s0 = gl_add_c(psd_sbox(s[0]), PSD_TAB(POSEIDON_PARTIAL_RC)[r]);
        gl_t d = gl_mul_small(s0, 17 + 8);   // MDS[0][0] = circ[0] + diag[0]
#pragma unroll
        for (int i = 1; i < 12; i++) d = gl_mul_add(d, s[i], PSD_TAB(POSEIDON_PARTIAL_ROW)[r * 11 + i - 1]);
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = gl_mul_add(s[i], s0, PSD_TAB(POSEIDON_PARTIAL_COL)[r * 11 + i - 1]);
        s[0] = d;
    }
}

GL_HD void psd_permute(gl_t (&s)[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) psd_full_round(s, r);
    psd_partial_rounds(s);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) psd_full_round(s, 4 + POSEIDON_PARTIAL_ROUNDS + r);
}

// two_to_one (hashing.rs:98-115): permute([l, r, 0,0,0,0])[0..4]
GL_HD void psd_two_to_one(const gl_t* l, const gl_t* r, gl_t* out) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = l[i]; s[4 + i] = r[i]; s[8 + i] = 0; }
    psd_permute(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = gl_canon(s[i]);
}
