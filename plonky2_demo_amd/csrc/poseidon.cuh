// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds), device + host.
// Same function as plonky2/src/hash/poseidon.rs:598-609 (pinned by the four known-answer vectors of
// plonky2/src/hash/poseidon_goldilocks.rs:449-485); the sponge / compression wrappers follow
// plonky2/src/hash/hashing.rs:98-146 and plonky2/src/plonk/config.rs:55-66.
//
// GPU (psd_permute under __HIP_DEVICE_COMPILE__): one lane owns one 12-word state (24 VGPRs) and runs the textbook round
// structure; the MDS layer exploits the 6-bit circulant entries: every state word is split into 32-bit halves and the two
// 12-term dot products are accumulated in 64-bit registers without intermediate reduction (v_mad_u64_u32), followed by
// one 96-bit reduction per output word -- the same lazy-reduction idea as poseidon.rs:176-198.
// Host (Challenger, public_inputs_hash, witness rows, verifier): the factorised "fast" partial rounds of
// poseidon.rs:311-366,399-427 with 128-bit lazy dot products; its tables are derived by tools/gen_poseidon_constants.py.
#pragma once
#include <stdlib.h>
#include "gl64.cuh"
#include "gl64_gfx950.cuh"
#include "poseidon_constants.h"

#if defined(__HIPCC__)
// device copies of the tables (per translation unit, constant address space -> scalar loads)
#define POSEIDON_TABLE(name, n) static __constant__ const uint64_t d_##name[n]
#define POSEIDON_TABLE32(name, n) static __constant__ const uint32_t d_##name[n]
#include "poseidon_constants.inc"
#undef POSEIDON_TABLE
#undef POSEIDON_TABLE32
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PSD_TAB(name) d_##name
#else
#define PSD_TAB(name) name
#endif

GL_HD gl_t psd_sbox(gl_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const gl_t x2 = glx_mul<false>(x, x), x4 = glx_mul<false>(x2, x2), x3 = glx_mul<false>(x, x2);
    return glx_mul<false>(x3, x4);
#else
    gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
#endif
}

// Host formulation (the Challenger, public_inputs_hash, the witness generator's sponge rows, the verifier): 64 x 64 -> 128
// multiplies are native on the CPU, so sums of products are accumulated unreduced in two 128-bit words (the constant is
// split into 32-bit halves: 12 terms of 64 x 32 bits stay below 2^100) and reduced once -- the dependent chain of a dot
// product is one reduction instead of twelve.
typedef unsigned __int128 psd_u128;
struct PsdHostDot { psd_u128 lo = 0, hi = 0; };
inline void psd_host_dot_term(PsdHostDot& d, gl_t s, gl_t c) { d.lo += (psd_u128)s * (uint32_t)c; d.hi += (psd_u128)s * (uint32_t)(c >> 32); }
inline gl_t psd_host_dot_reduce(const PsdHostDot& d) {
    const gl_t h = gl_reduce128((gl_t)d.hi, (gl_t)(d.hi >> 64));            // hi * 2^32
    const gl_t h32 = gl_reduce128(h << 32, h >> 32);
    return gl_add(gl_reduce128((gl_t)d.lo, (gl_t)(d.lo >> 64)), h32);
}
// MDS layer, host formulation (the GPU uses psd_mds_then_constants): the state's 32-bit halves are laid out twice in a
// row so that output r reads a contiguous window -- twelve independent 64-bit accumulators per half, which the compiler
// vectorises (SSE2) and which an AVX2 build of the same loop, selected at run time, does in 64 ns instead of 166.
GL_HD void psd_mds_finish(gl_t (&s)[12], const gl_t* al, const gl_t* ah) {
    for (int r = 0; r < 12; r++) {
        const gl_t l = al[r] + (ah[r] << 32);
        const uint32_t top = (uint32_t)(ah[r] >> 32) + ((l < al[r]) ? 1u : 0u);
        s[r] = gl_reduce96(l, top);
    }
}
GL_HD void psd_mds_portable(gl_t (&s)[12]) {
    const uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    gl_t lo2[24], hi2[24], al[12], ah[12];
    for (int i = 0; i < 12; i++) { lo2[i] = lo2[i + 12] = (uint32_t)s[i]; hi2[i] = hi2[i + 12] = s[i] >> 32; al[i] = 0; ah[i] = 0; }
    for (int i = 0; i < 12; i++) {
        const gl_t c = circ[i];
        for (int r = 0; r < 12; r++) { al[r] += lo2[i + r] * c; ah[r] += hi2[i + r] * c; }    // each sum < 2^41
    }
    al[0] += lo2[0] * 8; ah[0] += hi2[0] * 8;                                                    // MDS_MATRIX_DIAG[0] = 8
    psd_mds_finish(s, al, ah);
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) inline void psd_mds_avx2(gl_t (&s)[12]) {
    const uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    gl_t lo2[24], hi2[24];
    for (int i = 0; i < 12; i++) { lo2[i] = lo2[i + 12] = (uint32_t)s[i]; hi2[i] = hi2[i + 12] = s[i] >> 32; }
    __m256i al[3], ah[3];
    for (int k = 0; k < 3; k++) { al[k] = _mm256_setzero_si256(); ah[k] = _mm256_setzero_si256(); }
    for (int i = 0; i < 12; i++) {
        const __m256i c = _mm256_set1_epi64x(circ[i]);
        for (int k = 0; k < 3; k++) {
            al[k] = _mm256_add_epi64(al[k], _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i + 4 * k)), c));
            ah[k] = _mm256_add_epi64(ah[k], _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i + 4 * k)), c));
        }
    }
    gl_t AL[12], AH[12];
    for (int k = 0; k < 3; k++) { _mm256_storeu_si256((__m256i*)(AL + 4 * k), al[k]); _mm256_storeu_si256((__m256i*)(AH + 4 * k), ah[k]); }
    AL[0] += lo2[0] * 8; AH[0] += hi2[0] * 8;
    psd_mds_finish(s, AL, AH);
}
inline void psd_mds(gl_t (&s)[12]) {
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && !getenv("GL_HOST_MDS_PORTABLE");
    if (have_avx2) psd_mds_avx2(s); else psd_mds_portable(s);
}
#else
GL_HD void psd_mds(gl_t (&s)[12]) { psd_mds_portable(s); }
#endif

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
        for (int c = 1; c < 12; c++) {
            PsdHostDot d;
            for (int r = 1; r < 12; r++) psd_host_dot_term(d, s[r], POSEIDON_PARTIAL_INIT[(r - 1) * 11 + (c - 1)]);
            t[c] = psd_host_dot_reduce(d);
        }
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        const gl_t s0 = gl_add_c(psd_sbox(s[0]), POSEIDON_PARTIAL_RC[r]);
        PsdHostDot d;
        psd_host_dot_term(d, s0, 17 + 8);    // MDS[0][0] = circ[0] + diag[0]
        for (int i = 1; i < 12; i++) psd_host_dot_term(d, s[i], POSEIDON_PARTIAL_ROW[r * 11 + i - 1]);
        for (int i = 1; i < 12; i++) s[i] = gl_mul_add(s[i], s0, POSEIDON_PARTIAL_COL[r * 11 + i - 1]);
        s[0] = psd_host_dot_reduce(d);
    }
}

// al + ah 2^32 (al, ah < 2^63) reduced to one word
GL_HD gl_t psd_acc_reduce(gl_t al, gl_t ah) {
#if defined(__HIP_DEVICE_COMPILE__)
    return glx_acc_reduce(al, ah);      // carry add, carry add, multiply-add by EPS with carry, select, add: 5 instructions
#else
    const uint32_t al_hi = (uint32_t)(al >> 32), ah_lo = (uint32_t)ah;
    const uint32_t mid = al_hi + ah_lo;
    const uint32_t top = (uint32_t)(ah >> 32) + (mid < ah_lo ? 1u : 0u);
    return gl_reduce96(((gl_t)mid << 32) | (uint32_t)al, top);
#endif
}

// GPU formulation of the permutation: the TEXTBOOK round structure (poseidon.rs:573-596 `poseidon_naive`: every round is
// constants, S-box, full MDS), not the factorised partial rounds.  On gfx950 a 64 x 64 modular multiply costs ~26 VALU
// instructions while the MDS layer, whose entries are 6-bit constants, is 288 single-instruction multiply-adds plus 12
// reductions: 22 partial rounds as "one S-box + MDS" are ~15 % fewer instructions than "one S-box + 11 modular
// multiply-adds + an 11-term dot product" plus the dense 11 x 11 pre-matrix of the factorised form.  The next round's
// constants ride along as the addend of the first multiply-add of each accumulator, so the constant layer is free.
// Same function (pinned by the reference's known-answer vectors and by fast == naive in the oracle).
template <bool WITH_RC>
GL_HD void psd_mds_then_constants(gl_t (&s)[12], const gl_t* __restrict__ rc) {
    // the power-of-two entries are made opaque: otherwise the compiler turns x * 16 into a 64-bit shift-add, which needs
    // the 32-bit operand zero-extended into a register pair first (two extra moves per term); a multiply-add does not
    uint32_t k16 = 16, k2 = 2, k8 = 8;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(k16), "+s"(k2), "+s"(k8));
#endif
    const uint32_t circ[12] = {17, 15, 41, k16, k2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
#if defined(__HIP_DEVICE_COMPILE__)
    // three output words at a time: their accumulators are reduced together (glx_acc_reduce3: no wait states)
#pragma unroll
    for (int r0 = 0; r0 < 12; r0 += 3) {
        gl_t al[3], ah[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int r = r0 + k;
            // each < (256 + 8 + 1) * 2^32 < 2^41; the round constant is the scalar addend of the first multiply-add (glx_mad_k)
            if (WITH_RC) { const gl_t c = rc[r]; al[k] = glx_mad_k(lo[r], 17u, (uint64_t)(uint32_t)c); ah[k] = glx_mad_k(hi[r], 17u, c >> 32); }
            else { al[k] = (gl_t)lo[r] * 17u; ah[k] = (gl_t)hi[r] * 17u; }
#pragma unroll
            for (int i = 1; i < 12; i++) {
                al[k] += (gl_t)lo[(i + r) % 12] * circ[i];
                ah[k] += (gl_t)hi[(i + r) % 12] * circ[i];
            }
            if (r == 0) { al[k] += (gl_t)lo[0] * k8; ah[k] += (gl_t)hi[0] * k8; }   // MDS_MATRIX_DIAG[0] = 8
        }
        glx_acc_reduce3(al[0], ah[0], al[1], ah[1], al[2], ah[2], s[r0], s[r0 + 1], s[r0 + 2]);
    }
#else
#pragma unroll
    for (int r = 0; r < 12; r++) {
        gl_t al = 0, ah = 0;   // each < (256 + 8 + 1) * 2^32 < 2^41
        if (WITH_RC) { const gl_t c = rc[r]; al = (uint32_t)c; ah = c >> 32; }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (gl_t)lo[(i + r) % 12] * circ[i];
            ah += (gl_t)hi[(i + r) % 12] * circ[i];
        }
        if (r == 0) { al += (gl_t)lo[0] * k8; ah += (gl_t)hi[0] * k8; }   // MDS_MATRIX_DIAG[0] = 8
        s[r] = psd_acc_reduce(al, ah);
    }
#endif
}
// M[r][c] of the MDS matrix (circulant + diag(8, 0, ...)), compile-time after unrolling
GL_HD constexpr uint32_t psd_mds_entry(int r, int c) {
    constexpr uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return circ[(c - r + 12) % 12] + ((r == 0 && c == 0) ? 8u : 0u);
}
GL_HD void psd_sbox_all(gl_t (&s)[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // x^7 = (x * x^2) * x^4, three words at a time: glx_mul3 interleaves three independent products so that its carry chains
    // need no wait states
#pragma unroll
    for (int k = 0; k < 12; k += 3) {
        gl_t a2, b2, c2, a4, b4, c4, a3, b3, c3;
        glx_mul3<false>(s[k], s[k], s[k + 1], s[k + 1], s[k + 2], s[k + 2], a2, b2, c2);
        glx_mul3<false>(a2, a2, b2, b2, c2, c2, a4, b4, c4);
        glx_mul3<false>(s[k], a2, s[k + 1], b2, s[k + 2], c2, a3, b3, c3);
        glx_mul3<false>(a3, a4, b3, b4, c3, c4, s[k], s[k + 1], s[k + 2]);
    }
#else
    for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
#endif
}
#if defined(__HIPCC__)
// Three partial rounds at once (tools/gen_poseidon_constants.py derive_groups): with d_j = sbox(x_j) - a_j the change of lane
// 0 in round j of the group (a_j = lane 0 entering its S-box), everything else is linear, so lane 0 of the next two rounds
// needs only row 0 of M and of M^2 applied to the group's input state, and the state after the group is M^3 (entries
// < 2^25: still one multiply-add per 32-bit half) applied to it plus three rank-one corrections: ~910 instead of 3 x 500
// instructions.  `s` enters and leaves with the round constants of its next round already added.
// SUBST = false: x_j = a_j (the permutation).  SUBST = true: x_j = in[j], the PoseidonGate's S-box input wires, and a[j]
// returns the computed a_j for the constraint a_j - in[j] (gates/poseidon.rs:165-189).
template <bool SUBST>
__device__ __forceinline__ void psd_partial_group(gl_t (&s)[12], int g, const gl_t* in, gl_t* a) {
    const gl_t* __restrict__ K = d_POSEIDON_G3_K + 14 * g;
    // the matrix constants are re-read (scalar loads) in every group: hoisted out of the loop they exceed the SGPR file
    // and get parked in VGPR lanes (250 v_readlane per group)
    uint32_t z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(z));
#endif
    const uint32_t* __restrict__ R2 = d_POSEIDON_G3_R2 + z;
    const uint32_t* __restrict__ M3 = d_POSEIDON_G3_M3 + z;
    const uint32_t* __restrict__ V1 = d_POSEIDON_G3_V1 + z;
    const uint32_t* __restrict__ V2 = d_POSEIDON_G3_V2 + z;
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
    a[0] = s[0];
    const gl_t d0 = gl_sub(psd_sbox(SUBST ? in[0] : s[0]), s[0]);
    const uint32_t d0l = (uint32_t)d0, d0h = (uint32_t)(d0 >> 32);
    gl_t al = glx_mad_k(d0l, 25u, (uint64_t)(uint32_t)K[0]), ah = glx_mad_k(d0h, 25u, K[0] >> 32);      // M[0][0]
#pragma unroll
    for (int i = 0; i < 12; i++) { al = glx_mac_c(al, lo[i], psd_mds_entry(0, i)); ah = glx_mac_c(ah, hi[i], psd_mds_entry(0, i)); }
    const gl_t a1 = psd_acc_reduce(al, ah);
    a[1] = a1;
    const gl_t d1 = gl_sub(psd_sbox(SUBST ? in[1] : a1), a1);
    const uint32_t d1l = (uint32_t)d1, d1h = (uint32_t)(d1 >> 32);
    al = glx_mad_k(d1l, 25u, (uint64_t)(uint32_t)K[1]); ah = glx_mad_k(d1h, 25u, K[1] >> 32);      // (inline-constant coefficient first)
#pragma unroll
    for (int i = 0; i < 12; i++) { const uint32_t c = R2[i]; al += (gl_t)lo[i] * c; ah += (gl_t)hi[i] * c; }
    { const uint32_t c = V1[0]; al += (gl_t)d0l * c; ah += (gl_t)d0h * c; }
    const gl_t a2 = psd_acc_reduce(al, ah);
    a[2] = a2;
    const gl_t d2 = gl_sub(psd_sbox(SUBST ? in[2] : a2), a2);
    const uint32_t d2l = (uint32_t)d2, d2h = (uint32_t)(d2 >> 32);
#pragma unroll
    for (int l0 = 0; l0 < 12; l0 += 3) {
        gl_t xl[3], xh[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int l = l0 + k;
            // the term with a compile-time (inline-constant) coefficient goes first: its multiply-add takes the round constant as
            // a scalar 64-bit addend, which a multiply-add with a scalar coefficient cannot (one scalar operand per instruction)
            // -- otherwise every accumulator starts with two register moves
            xl[k] = glx_mad_k(d2l, psd_mds_entry(l, 0), (uint64_t)(uint32_t)K[2 + l]); xh[k] = glx_mad_k(d2h, psd_mds_entry(l, 0), K[2 + l] >> 32);
#pragma unroll
            for (int i = 0; i < 12; i++) { const uint32_t c = M3[12 * l + i]; xl[k] += (gl_t)lo[i] * c; xh[k] += (gl_t)hi[i] * c; }
            { const uint32_t c = V2[l]; xl[k] += (gl_t)d0l * c; xh[k] += (gl_t)d0h * c; }
            { const uint32_t c = V1[l]; xl[k] += (gl_t)d1l * c; xh[k] += (gl_t)d1h * c; }
        }
        glx_acc_reduce3(xl[0], xh[0], xl[1], xh[1], xl[2], xh[2], s[l0], s[l0 + 1], s[l0 + 2]);
    }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
GL_HD void psd_permute(gl_t (&s)[12]) {       // (host + device only so that host code parses in the device pass)
    const gl_t* __restrict__ rc = d_POSEIDON_RC;
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], rc[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) { psd_sbox_all(s); psd_mds_then_constants<true>(s, rc + 12 * (r + 1)); }
#pragma unroll 1
    for (int g = 0; g < POSEIDON_PARTIAL_GROUPS; g++) { gl_t a[3]; psd_partial_group<false>(s, g, nullptr, a); }
#pragma unroll 1
    for (int r = 4 + 3 * POSEIDON_PARTIAL_GROUPS; r < 4 + POSEIDON_PARTIAL_ROUNDS; r++) { s[0] = psd_sbox(s[0]); psd_mds_then_constants<true>(s, rc + 12 * (r + 1)); }
#pragma unroll 1
    for (int r = 4 + POSEIDON_PARTIAL_ROUNDS; r < 7 + POSEIDON_PARTIAL_ROUNDS; r++) { psd_sbox_all(s); psd_mds_then_constants<true>(s, rc + 12 * (r + 1)); }
    psd_sbox_all(s);
    psd_mds_then_constants<false>(s, nullptr);
}
#else
GL_HD void psd_permute(gl_t (&s)[12]) {         // host formulation (GL_HD only so that kernels parse in the host pass)
    for (int r = 0; r < 4; r++) psd_full_round(s, r);
    psd_partial_rounds(s);
    for (int r = 0; r < 4; r++) psd_full_round(s, 4 + POSEIDON_PARTIAL_ROUNDS + r);
}
#endif

#if defined(__HIPCC__)
// Cooperative permutation for LATENCY-bound launches (the upper levels of a Merkle tree, the small FRI trees): 16 lanes
// share one state, lane l < 12 holds word l, so the 12 S-boxes of a full round run side by side and the MDS layer is 22
// cross-lane reads (ds_bpermute) plus 24 multiply-adds per lane.  The dependent chain of a permutation shrinks from
// ~23 k to ~5 k instructions; three quarters of the lanes' issue slots are idle, so it only pays when the launch could
// not fill the chip anyway.  `l` = lane & 15; lanes 12..15 carry zeros and mirror word 0's control flow.
__device__ __forceinline__ gl_t psd_coop_permute(gl_t s, const int l) {
    const uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const gl_t* __restrict__ rc = d_POSEIDON_RC;
    const int lc = l < 12 ? l : 0;
    const int lane = (int)(threadIdx.x & 63), row = lane & ~15;
    s = gl_add_c(s, rc[lc]);
#pragma unroll 1
    for (int r = 0; r < 8 + POSEIDON_PARTIAL_ROUNDS; r++) {
        const bool full = r < 4 || r >= 4 + POSEIDON_PARTIAL_ROUNDS;
        if (full || l == 0) s = psd_sbox(s);
        const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);
        gl_t al = 0, ah = 0;
        if (r + 1 < 8 + POSEIDON_PARTIAL_ROUNDS) { const gl_t c = rc[12 * (r + 1) + lc]; al = (uint32_t)c; ah = c >> 32; }
        al = glx_mac_c(al, lo, 17u); ah = glx_mac_c(ah, hi, 17u);      // circ[0]; multiply-adds, not shift-adds (glx_mac_c)
#pragma unroll
        for (int i = 1; i < 12; i++) {
            int src = lc + i; src -= (src >= 12) ? 12 : 0;
            const int addr = (row + src) << 2;
            const uint32_t lo_i = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)lo);
            const uint32_t hi_i = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)hi);
            al = glx_mac_c(al, lo_i, circ[i]); ah = glx_mac_c(ah, hi_i, circ[i]);
        }
        if (l == 0) { al += (gl_t)lo * 8; ah += (gl_t)hi * 8; }   // MDS_MATRIX_DIAG[0] = 8
        s = psd_acc_reduce(al, ah);
        if (l >= 12) s = 0;
    }
    return s;
}
#endif

// two_to_one (hashing.rs:98-115): permute([l, r, 0,0,0,0])[0..4]
GL_HD void psd_two_to_one(const gl_t* l, const gl_t* r, gl_t* out) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = l[i]; s[4 + i] = r[i]; s[8 + i] = 0; }
    psd_permute(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = gl_canon(s[i]);
}
