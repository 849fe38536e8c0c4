// Device kernels for the prove() phases between the polynomial commitments.
//   k_pp_chunk_products / k_z_*      permutation argument: quotient chunk products, running product Z and partial
//                                    products   (plonk/prover.rs:359-416, util/partial_products.rs:13-37)
//   k_quotient                       vanishing-polynomial evaluation on the LDE coset, x 1/Z_H
//                                    (plonk/prover.rs:576-737, plonk/vanishing_poly.rs:164-330, gates/*.rs)
//   k_eval_at_ext                    OpeningSet::new: p(zeta) for every committed polynomial (plonk/proof.rs:306-344)
//   k_fri_combine / k_div_linear_*   prove_openings: sum_j alpha^j f_j and (F - F(z))/(X - z)
//                                    (fri/oracle.rs:162-219, util/reducing.rs:83-106, polynomial/division.rs:75-88)
//   k_fri_fold                       commit-phase coefficient fold (fri/prover.rs:94-103)
//   k_pow_grind                      proof of work (fri/prover.rs:115-160), minimum witness
//   k_gather_*                       query answers (fri/prover.rs:162-216)
// All values are exact field elements; every reordering used here (per-gate alpha sums, forward alpha powers,
// chunk-level inversion, segmented scans) is an identity in F_p / F_p^2, so results equal the reference's.
// Data layout: column-major "structure of polynomials" ([poly][index]); extension polynomials as two planes.
#pragma once
#include "gl64.cuh"
#include "poseidon.cuh"
#include "../../include/plonky2_mi355x.h"

#define GLP_MAX_ROUTED 80
#define GLP_CHUNKS 10           // ceil(80 / 8)

struct GlPermParams {
    const gl_t* wires;          // witness VALUES [num_wires][n]
    const gl_t* sigmas;         // sigma VALUES [80][n]
    const gl_t* xpow_lo; const gl_t* xpow_hi;   // two-level table of w_n^i
    gl_t k_is[GLP_MAX_ROUTED];
    gl_t betas[2], gammas[2];
    uint32_t n;
    uint32_t k_is_powers_of_7;  // k_is[j] = 7^j (get_unique_coset_shifts, field/src/cosets.rs:9-24): beta x k_j by a running x7
    gl_t* chunk_prod;           // [2][10][n]
    gl_t* row_prod;             // [2][n]
};

__device__ __forceinline__ gl_t glp_pow2level(const gl_t* lo, const gl_t* hi, uint32_t e) {
    return gl_mul(lo[e & 2047], hi[e >> 11]);
}

// The 10 chunk products  prod_{j in chunk} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)  of every row, in two
// launches so that the long dependent multiply chain is spread over 20 x more lanes:
//   k_pp_chunk_terms     one thread per (row, challenge, chunk): numerator and denominator product of the chunk's 8 wires
//                        (grid.y = 2 * 10); the numerator goes to chunk_prod, the denominator to den_prod
//   k_pp_chunk_products  one thread per (row, challenge): ONE inversion for the 10 denominators (Montgomery), the quotients and
//                        their product over the row
__global__ __launch_bounds__(256) void k_pp_chunk_terms(GlPermParams p, gl_t* __restrict__ den_prod /* [2][10][n] */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const int a = blockIdx.y / GLP_CHUNKS, c = blockIdx.y % GLP_CHUNKS;
    const gl_t x = glp_pow2level(p.xpow_lo, p.xpow_hi, i);
    const gl_t beta = p.betas[a], gamma = p.gammas[a];
    const gl_t bx = gl_mul(beta, x);
    gl_t np = 1, dp = 1;
    gl_t bxk = gl_mul(bx, p.k_is[c * 8]);               // beta x k_j, advanced by x7 when the shifts are the usual 7^j
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int j = c * 8 + q;
        const gl_t w = p.wires[(size_t)j * p.n + i];
        const gl_t s = p.sigmas[(size_t)j * p.n + i];
        if (q && !p.k_is_powers_of_7) bxk = gl_mul(bx, p.k_is[j]);
        np = gl_mul(np, gl_add(gl_add(w, bxk), gamma));
        dp = gl_mul(dp, gl_add(gl_mul_add(w, beta, s), gamma));
        if (p.k_is_powers_of_7) bxk = gl_mul_small(bxk, 7);
    }
    p.chunk_prod[((size_t)a * GLP_CHUNKS + c) * p.n + i] = np;
    den_prod[((size_t)a * GLP_CHUNKS + c) * p.n + i] = dp;
}
__global__ __launch_bounds__(256) void k_pp_chunk_products(GlPermParams p, const gl_t* __restrict__ den_prod) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const int a = blockIdx.y;
    gl_t nump[GLP_CHUNKS], denp[GLP_CHUNKS];
#pragma unroll
    for (int c = 0; c < GLP_CHUNKS; c++) {
        nump[c] = p.chunk_prod[((size_t)a * GLP_CHUNKS + c) * p.n + i];
        denp[c] = den_prod[((size_t)a * GLP_CHUNKS + c) * p.n + i];
    }
    gl_t pre[GLP_CHUNKS];
    gl_t acc = 1;
#pragma unroll
    for (int c = 0; c < GLP_CHUNKS; c++) { pre[c] = acc; acc = gl_mul(acc, denp[c]); }
    gl_t inv = gl_inv(acc);
    gl_t rowp = 1;
#pragma unroll
    for (int c = GLP_CHUNKS - 1; c >= 0; c--) {
        gl_t dinv = gl_mul(inv, pre[c]);
        inv = gl_mul(inv, denp[c]);
        nump[c] = gl_mul(nump[c], dinv);
    }
#pragma unroll
    for (int c = 0; c < GLP_CHUNKS; c++) {
        p.chunk_prod[((size_t)a * GLP_CHUNKS + c) * p.n + i] = gl_canon(nump[c]);
        rowp = gl_mul(rowp, nump[c]);
    }
    p.row_prod[(size_t)a * p.n + i] = gl_canon(rowp);
}

// ---- exclusive multiplicative scan of row_prod over rows: Z(x_i) = prod_{i' < i} P_i' -----------------------
#define GLP_SEG 2048            // rows per workgroup segment (256 threads x 8)

// S1: product of each segment
__global__ __launch_bounds__(256) void k_z_segment_products(const gl_t* row_prod, uint32_t n, gl_t* seg_prod) {
    __shared__ gl_t sh[256];
    const uint32_t a = blockIdx.y, seg = blockIdx.x, t = threadIdx.x;
    const gl_t* rp = row_prod + (size_t)a * n;
    gl_t acc = 1;
#pragma unroll
    for (int q = 0; q < 8; q++) { uint32_t i = seg * GLP_SEG + t * 8 + q; if (i < n) acc = gl_mul(acc, rp[i]); }
    sh[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sh[t] = gl_mul(sh[t], sh[t + s]); __syncthreads(); }
    if (t == 0) seg_prod[(size_t)a * gridDim.x + seg] = gl_canon(sh[0]);
}
// S2: exclusive scan of the segment products (one thread per challenge; segment counts are tiny)
__global__ void k_z_segment_scan(gl_t* seg_prod, uint32_t nseg) {
    const uint32_t a = threadIdx.x;
    if (a >= 2) return;
    gl_t acc = 1;
    for (uint32_t s = 0; s < nseg; s++) { gl_t v = seg_prod[(size_t)a * nseg + s]; seg_prod[(size_t)a * nseg + s] = acc; acc = gl_canon(gl_mul(acc, v)); }
}
// S3: per-row Z and the nine partial products.  out columns: [Z_0, Z_1, pp_0[0..9), pp_1[0..9)], each n values
__global__ __launch_bounds__(256) void k_z_finalize(const gl_t* row_prod, const gl_t* chunk_prod, const gl_t* seg_excl, uint32_t n, gl_t* out) {
    __shared__ gl_t sh[256];
    const uint32_t a = blockIdx.y, seg = blockIdx.x, t = threadIdx.x, nseg = gridDim.x;
    const gl_t* rp = row_prod + (size_t)a * n;
    gl_t loc[8];
    gl_t acc = 1;
#pragma unroll
    for (int q = 0; q < 8; q++) { uint32_t i = seg * GLP_SEG + t * 8 + q; loc[q] = acc; if (i < n) acc = gl_mul(acc, rp[i]); }
    sh[t] = acc;
    __syncthreads();
    // exclusive scan of the 256 per-thread products (Hillis-Steele on inclusive values)
    for (int d = 1; d < 256; d <<= 1) {
        gl_t v = (t >= (uint32_t)d) ? sh[t - d] : 1;
        __syncthreads();
        sh[t] = gl_mul(sh[t], v);
        __syncthreads();
    }
    gl_t prefix = gl_mul(seg_excl[(size_t)a * nseg + seg], t ? sh[t - 1] : 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        uint32_t i = seg * GLP_SEG + t * 8 + q;
        if (i >= n) break;
        gl_t z = gl_mul(prefix, loc[q]);
        out[(size_t)a * n + i] = gl_canon(z);                                     // Z_a(x_i)
        gl_t run = z;
#pragma unroll
        for (int c = 0; c < GLP_CHUNKS - 1; c++) {
            run = gl_mul(run, chunk_prod[((size_t)a * GLP_CHUNKS + c) * n + i]);
            out[(size_t)(2 + a * (GLP_CHUNKS - 1) + c) * n + i] = gl_canon(run);  // partial product c of challenge a
        }
    }
}

// ---- quotient ------------------------------------------------------------------------------------------------
struct GlQuotParams {
    const gl_t* cs;             // constants||sigmas LDE [num_constants + 80][N] natural order
    const gl_t* wires;          // wires LDE [135][N]
    const gl_t* zs;             // Z||partial products(||lookup polynomials) LDE [20 (+ 14)][N]
    const gl_t* xpow_lo; const gl_t* xpow_hi;    // 7 * w_N^i two-level
    const gl_t* alpha_pows;     // [2][GLQ_MAX_TERMS]: alpha_b^t
    gl_t* out;                  // [2][N]
    gl_t k_is[GLP_MAX_ROUTED];
    gl_t betas[2], gammas[2];
    gl_t pi_hash[4];
    gl_t zh_evals[8], zh_inv[8];    // Z_H on the coset by i mod 8 and inverses (field/src/zero_poly_coset.rs)
    gl_t n_field;                   // n as a field element
    const gl_t* l0_coset;           // L_0 on the coset: l0_coset[i] = Z_H(x_i) / (n (x_i - 1)), x_i = 7 w_N^i (built with the circuit)
    uint32_t lgN, num_constants, num_selectors, num_gates, next_step;
    uint32_t k_is_powers_of_7;
    uint8_t gate_types[GL_MAX_GATES], gate_params[GL_MAX_GATES];
    uint32_t gate_sel[GL_MAX_GATES], group_start[GL_MAX_GATES], group_end[GL_MAX_GATES];
    // lookup argument (zero without lookups): the lookup selector columns sit between the gate selectors and the gates' constants, the
    // lookup polynomials behind Z and the partial products, their 2 x (16 + num_luts) terms between the partial-product checks and the
    // gate terms
    uint32_t num_lookup_selectors, num_lookup_polys, num_luts, gate_term0;
    gl_t deltas[8];                 // per challenge: ChallengeA, ChallengeB, ChallengeAlpha, ChallengeDelta (circuit_builder.rs:61-71)
    gl_t lut_poly_at_delta[2][GL_MAX_LUTS];      // get_lut_poly(table).eval(delta) per challenge (vanishing_poly.rs:31-49), computed on the host
};
#define GLQ_MAX_TERMS 192
// per challenge: last LDC, initial Sum, initial RE, final RE (one per table), RE transition, 6 x (Sum, LDC) transitions
#define GLQ_LOOKUP_TERMS(num_luts) (16u + (num_luts))

// running alpha-weighted sums for the two alphas: unreduced (GlxWideAcc2: 16 instructions per term for both), one reduction
// when the sum is used
struct GlAlphaAcc {
    GlxWideAcc2 w;
    const gl_t* ap;             // alpha_pows base (uniform: the weights are scalar operands)
    __device__ __forceinline__ void start(const gl_t* alpha_pows) { w.clear(); ap = alpha_pows; }
    __device__ __forceinline__ void add(uint32_t t, gl_t term) { w.mac(term, ap[t], ap[GLQ_MAX_TERMS + t]); }
    __device__ __forceinline__ gl_t sum(int b) const { return w.sum(b); }
};

// PoseidonGate constraints (gates/poseidon.rs:193-272), term index base `t0`.  Wire k of the point lives at w[k * N]; the
// wires are read through WALKING pointers (p += N) in the order the gate consumes them: with closed-form addresses the
// compiler hoists all 135 loop-invariant 64-bit addresses out of the gate loop and the kernel needs ~290 VGPRs.
// Textbook round structure (constants, S-box, MDS) with the next round's constants folded into the MDS accumulators
// (psd_mds_then_constants): the wire holding an S-box input is compared with the state right after the constants, which is
// the same value in every factorisation of the partial rounds (gates/poseidon.rs:139-189).
// advance a wire pointer and hide the result from the optimiser, so that the addresses of later wires cannot be formed
// (and kept in registers) ahead of time
__device__ __forceinline__ const gl_t* glq_step(const gl_t* p, size_t stride) {
    p += stride;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ void glq_poseidon_gate(const gl_t* __restrict__ w, size_t N, GlAlphaAcc& acc, uint32_t t0) {
    uint32_t t = t0;
    gl_t s[12];
    {
        const gl_t* wp = w;                              // wires 0..11: inputs
#pragma unroll
        for (int i = 0; i < 12; i++) { s[i] = *wp; wp = glq_step(wp, N); }
        wp = glq_step(wp, 12 * N);                                    // skip the outputs (12..23)
        const gl_t swap = *wp; wp = glq_step(wp, N);                  // 24
        acc.add(t++, gl_mul(swap, gl_sub(swap, 1)));
#pragma unroll
        for (int i = 0; i < 4; i++) {                    // 25..28: delta_i = swap * (rhs - lhs)
            const gl_t delta = *wp; wp = glq_step(wp, N);
            const gl_t lhs = s[i], rhs = s[i + 4];
            acc.add(t++, gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
            s[i] = gl_add(lhs, delta);
            s[i + 4] = gl_sub(rhs, delta);
        }
    }
    const gl_t* __restrict__ rc = d_POSEIDON_RC;
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], rc[i]);
    const gl_t* wp = w + 29 * N;                         // 29..64: S-box inputs of full rounds 1..3
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) { const gl_t in = *wp; wp = glq_step(wp, N); acc.add(t++, gl_sub(s[i], in)); s[i] = in; }
        }
        psd_sbox_all(s);
        psd_mds_then_constants<true>(s, rc + 12 * (r + 1));
    }
#pragma unroll 1
    for (int g = 0; g < POSEIDON_PARTIAL_GROUPS; g++) {  // 65..85: three partial rounds at a time (psd_partial_group)
        gl_t in[3], a[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { in[j] = *wp; wp = glq_step(wp, N); }
        psd_partial_group<true>(s, g, in, a);
#pragma unroll
        for (int j = 0; j < 3; j++) acc.add(t++, gl_sub(a[j], in[j]));
    }
#pragma unroll 1
    for (int r = 3 * POSEIDON_PARTIAL_GROUPS; r < POSEIDON_PARTIAL_ROUNDS; r++) {  // 86
        const gl_t in = *wp; wp = glq_step(wp, N);
        acc.add(t++, gl_sub(s[0], in));
        s[0] = psd_sbox(in);
        psd_mds_then_constants<true>(s, rc + 12 * (4 + r + 1));
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {                        // 87..134
#pragma unroll
        for (int i = 0; i < 12; i++) { const gl_t in = *wp; wp = glq_step(wp, N); acc.add(t++, gl_sub(s[i], in)); s[i] = in; }
        psd_sbox_all(s);
        if (r < 3) psd_mds_then_constants<true>(s, rc + 12 * (4 + POSEIDON_PARTIAL_ROUNDS + r + 1));
        else psd_mds_then_constants<false>(s, nullptr);
    }
    wp = w + 12 * N;                                     // 12..23: outputs
#pragma unroll
    for (int i = 0; i < 12; i++) { acc.add(t++, gl_sub(s[i], *wp)); wp = glq_step(wp, N); }
}

// L_0(x) = Z_H(x) / (n (x - 1)) on the N coset points (plonk_common.rs:61-71, zero_poly_coset.rs:55-60): one field inversion per
// point, so it is tabulated once per circuit instead of being recomputed (96 multiplies) by every proof at every point
__global__ __launch_bounds__(256) void k_l0_on_coset(const gl_t* xpow_lo, const gl_t* xpow_hi, uint32_t N, gl_t n_field, const gl_t* zh_evals8, gl_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const gl_t x = glp_pow2level(xpow_lo, xpow_hi, i);
    out[i] = gl_canon(gl_mul(zh_evals8[i & 7], gl_inv(gl_mul(n_field, gl_sub(x, 1)))));
}

// The vanishing combination is a sum over terms, so it is evaluated by two launches with very different register needs
// (the PoseidonGate re-runs a permutation; everything else streams wires): POSEIDON_PART = false writes every term except
// the PoseidonGate's, POSEIDON_PART = true adds the PoseidonGate's filtered constraint sum to it.
// (at least 3 waves per SIMD: the PoseidonGate part otherwise takes 235 VGPRs for a 5 % slower kernel)
template <bool POSEIDON_PART>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_quotient(GlQuotParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N = size_t(1) << p.lgN;
    if (i >= N) return;
    const gl_t* w = p.wires + i;
    const gl_t* cs = p.cs + i;
    const gl_t* zs = p.zs + i;
    // every operand read below is canonical: LDE values and tables are written by the NTT / table kernels, which store
    // canonical words, and the challenges come from the transcript
    GlAlphaAcc terms; terms.start(p.alpha_pows);
    gl_t tot0 = 0, tot1 = 0;                                         // canonical running totals
    if constexpr (!POSEIDON_PART) {
    const gl_t x = glx_canon(glp_pow2level(p.xpow_lo, p.xpow_hi, i));          // 7 * w^i
    const uint32_t i_next = (i + p.next_step) & (uint32_t)(N - 1);
    // L_0(x) (Z(x) - 1)            (vanishing_poly.rs:263-268; zero_poly_coset.rs:55-60)
    const gl_t l0 = p.l0_coset[i];
    gl_t bx0, bx1, lz0, lz1, bxk0, bxk1, unused;
    glx_mul3<true>(p.betas[0], x, p.betas[1], x, l0, glx_sub_cc(zs[0], 1), bx0, bx1, lz0);
    terms.add(0, lz0);
    glx_mul3<true>(bx0, p.k_is[0], bx1, p.k_is[0], l0, glx_sub_cc(zs[N], 1), bxk0, bxk1, lz1);   // beta x k_j: a running x7 when k_j = 7^j
    terms.add(1, lz1);
    // partial-product checks (util/partial_products.rs:52-76): terms 2 + 10 a + c
    {
        const gl_t beta0 = p.betas[0], beta1 = p.betas[1], gamma0 = p.gammas[0], gamma1 = p.gammas[1];
        gl_t prev0 = zs[0], prev1 = zs[N];
#pragma unroll 1
        for (int c = 0; c < GLP_CHUNKS; c++) {
            gl_t n0 = 1, d0 = 1, n1 = 1, d1 = 1;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = c * 8 + q;
                const gl_t wv = w[(size_t)j * N], sg = cs[(size_t)(p.num_constants + j) * N];
                if (!p.k_is_powers_of_7) { const gl_t k = p.k_is[j]; glx_mul3<true>(bx0, k, bx1, k, 0, 0, bxk0, bxk1, unused); }
                // numerator factor w + beta k_j x + gamma, denominator factor w + beta sigma_j(x) + gamma; six products per
                // wire, three at a time (the carry chains of three independent products interleave)
                const gl_t f0 = glx_add_cc(glx_add_cc(wv, bxk0), gamma0), f1 = glx_add_cc(glx_add_cc(wv, bxk1), gamma1);
                gl_t sb0, sb1;
                glx_mul3<true>(sg, beta0, sg, beta1, n0, f0, sb0, sb1, n0);
                const gl_t g0 = glx_add_cc(glx_add_cc(wv, sb0), gamma0), g1 = glx_add_cc(glx_add_cc(wv, sb1), gamma1);
                glx_mul3<true>(d0, g0, n1, f1, d1, g1, d0, n1, d1);
                if (p.k_is_powers_of_7) {                             // 7 y = 8 y - y
                    bxk0 = glx_sub_cc(glx_shl_c<3>(bxk0), bxk0); bxk1 = glx_sub_cc(glx_shl_c<3>(bxk1), bxk1);
                }
            }
            const gl_t next0 = (c == GLP_CHUNKS - 1) ? p.zs[i_next] : zs[(size_t)(2 + c) * N];
            const gl_t next1 = (c == GLP_CHUNKS - 1) ? p.zs[N + i_next] : zs[(size_t)(2 + (GLP_CHUNKS - 1) + c) * N];
            gl_t pn0, nd0, pn1, nd1;
            glx_mul3<true>(prev0, n0, next0, d0, prev1, n1, pn0, nd0, pn1);
            nd1 = glx_mul<true>(next1, d1);
            terms.add(2 + c, glx_sub_cc(pn0, nd0));
            terms.add(2 + GLP_CHUNKS + c, glx_sub_cc(pn1, nd1));
            prev0 = next0; prev1 = next1;
        }
    }
    tot0 = terms.sum(0); tot1 = terms.sum(1);
    }
    // gate constraints: sum_g filter_g * sum_j alpha^(22+j) c_{g,j}   (vanishing_poly.rs:706-732, gate.rs:121-146)
    const uint32_t T0 = p.gate_term0;                               // 2 + 2 * GLP_CHUNKS, + 2 x 17 lookup terms when there are lookups
    const gl_t* gc = cs + (size_t)(p.num_selectors + p.num_lookup_selectors) * N;      // the gate's own constants (gate.rs:129-133)
#pragma unroll 1
    for (uint32_t g = 0; g < p.num_gates; g++) {
        if ((p.gate_types[g] == 4) != POSEIDON_PART) continue;
        const gl_t sel = cs[(size_t)p.gate_sel[g] * N];
        gl_t filter = 1;                                            // gate.rs:277-284
        for (uint32_t k = p.group_start[g]; k < p.group_end[g]; k++) if (k != g) filter = glx_mul<true>(filter, glx_sub_cc((gl_t)k, sel));
        if (p.num_selectors > 1) filter = glx_mul<true>(filter, glx_sub_cc((gl_t)0xFFFFFFFFull, sel));
        GlAlphaAcc acc; acc.start(p.alpha_pows);
        if constexpr (POSEIDON_PART) glq_poseidon_gate(w, N, acc, T0);
        else switch (p.gate_types[g]) {
            case 1:     // ConstantGate (gates/constant.rs:59-66)
                acc.add(T0, glx_sub_cc(gc[0], w[0]));
                acc.add(T0 + 1, glx_sub_cc(gc[N], w[N]));
                break;
            case 2:     // PublicInputGate (gates/public_input.rs:44-49)
#pragma unroll
                for (int k = 0; k < 4; k++) acc.add(T0 + k, glx_sub_cc(w[(size_t)k * N], p.pi_hash[k]));
                break;
            case 3: {   // ArithmeticGate (gates/arithmetic_base.rs:163-181): two operations at a time, six products in two groups
                const gl_t c0 = gc[0], c1 = gc[N];
#pragma unroll 2
                for (int k = 0; k < 20; k += 2) {
                    const gl_t* wk = w + (size_t)(4 * k) * N;
                    const gl_t m0a = wk[0], m1a = wk[N], ada = wk[2 * N], oa = wk[3 * N], m0b = wk[4 * N], m1b = wk[5 * N], adb = wk[6 * N], ob = wk[7 * N];
                    gl_t pa, pb, qa, qb;
                    glx_mul3<true>(m0a, m1a, m0b, m1b, ada, c1, pa, pb, qa);
                    glx_mul3<true>(pa, c0, pb, c0, adb, c1, pa, pb, qb);
                    acc.add(T0 + k, glx_sub_cc(oa, glx_add_cc(pa, qa)));
                    acc.add(T0 + k + 1, glx_sub_cc(ob, glx_add_cc(pb, qb)));
                }
                break;
            }
            case 5: {   // BaseSumGate<2>, 63 limbs (gates/base_sum.rs:153-170): sum of limb_i 2^i - sum, then limb (limb - 1) per limb
                gl_t computed = 0;                                   // reduce_with_powers(limbs, 2): Horner from the top limb
#pragma unroll 1
                for (int k = 63; k >= 1; k -= 3) {                   // wires 1..63 = limbs 0..62, three range checks per step
                    const gl_t l2 = w[(size_t)k * N], l1 = w[(size_t)(k - 1) * N], l0 = w[(size_t)(k - 2) * N];
                    computed = glx_add_cc(glx_add_cc(computed, computed), l2);
                    computed = glx_add_cc(glx_add_cc(computed, computed), l1);
                    computed = glx_add_cc(glx_add_cc(computed, computed), l0);
                    gl_t r2, r1, r0;
                    glx_mul3<true>(l2, glx_sub_cc(l2, 1), l1, glx_sub_cc(l1, 1), l0, glx_sub_cc(l0, 1), r2, r1, r0);
                    acc.add(T0 + k, r2); acc.add(T0 + k - 1, r1); acc.add(T0 + k - 2, r0);
                }
                acc.add(T0, glx_sub_cc(computed, w[0]));
                break;
            }
            case 8: {   // ExponentiationGate, 66 power bits (gates/exponentiation.rs:196-228): square-and-multiply, bits big-endian; wires:
                        // 0 base, 1..66 bits, 67 output, 68..133 intermediate values.  Three steps at a time (their six products in two groups).
                const gl_t base = w[0];
                const gl_t* iv = w + (size_t)68 * N;                 // intermediate value i at iv[i * N]
                gl_t last = 1;                                       // intermediate value i - 1 (1 before the first step)
#pragma unroll 1
                for (int k = 0; k < 66; k += 3) {
                    const gl_t v0 = iv[(size_t)k * N], v1 = iv[(size_t)(k + 1) * N], v2 = iv[(size_t)(k + 2) * N];
                    const gl_t b0 = w[(size_t)(66 - k) * N], b1 = w[(size_t)(65 - k) * N], b2 = w[(size_t)(64 - k) * N];      // bit wire 1 + (65 - i)
                    gl_t p0, p1, p2, m0, m1, m2;
                    glx_mul3<true>(last, last, v0, v0, v1, v1, p0, p1, p2);            // prev_i = (intermediate value i - 1)^2
                    if (k == 0) p0 = 1;
                    glx_mul3<true>(b0, base, b1, base, b2, base, m0, m1, m2);          // bit base + (1 - bit)
                    m0 = glx_add_cc(m0, glx_sub_cc(1, b0)); m1 = glx_add_cc(m1, glx_sub_cc(1, b1)); m2 = glx_add_cc(m2, glx_sub_cc(1, b2));
                    glx_mul3<true>(p0, m0, p1, m1, p2, m2, p0, p1, p2);
                    acc.add(T0 + k, glx_sub_cc(p0, v0)); acc.add(T0 + k + 1, glx_sub_cc(p1, v1)); acc.add(T0 + k + 2, glx_sub_cc(p2, v2));
                    last = v2;
                }
                acc.add(T0 + 66, glx_sub_cc(w[(size_t)67 * N], last));
                break;
            }
            default: break;   // NoopGate
        }
        gl_t fs0, fs1, unused;
        glx_mul3<true>(filter, acc.sum(0), filter, acc.sum(1), 0, 0, fs0, fs1, unused);
        tot0 = glx_add_cc(tot0, fs0); tot1 = glx_add_cc(tot1, fs1);
    }
    const gl_t zi = p.zh_inv[i & 7];
    gl_t o0, o1, unused;
    glx_mul3<true>(tot0, zi, tot1, zi, 0, 0, o0, o1, unused);
    if constexpr (POSEIDON_PART) {                                   // (the other launch wrote canonical words)
        p.out[i] = glx_add_cc(p.out[i], o0);
        p.out[N + i] = glx_add_cc(p.out[N + i], o1);
    } else {
        p.out[i] = o0;
        p.out[N + i] = o1;
    }
}

// ---- lookup argument: check_lookup_constraints_batch (plonk/vanishing_poly.rs:503-670) on the LDE points -----------------------------
// One thread per point, both challenges; ADDS sum_t alpha^t term_t / Z_H(x) to what k_quotient wrote (the combination is linear).
// Only circuits with a lookup table launch it; every operand is canonical.
__global__ __launch_bounds__(256) void k_quotient_lookup(GlQuotParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N = size_t(1) << p.lgN;
    if (i >= N) return;
    const gl_t* w = p.wires + i;
    const gl_t* sel = p.cs + i + (size_t)p.num_selectors * N;       // lookup selectors: TransSre, TransLdc, InitSre, LastLdc, one end per table
    const uint32_t i_next = (i + p.next_step) & (uint32_t)(N - 1);
    const uint32_t NLP = p.num_lookup_polys, num_sldc = NLP - 1;    // 7 and 6
    const gl_t s_trans_sre = sel[0], s_trans_ldc = sel[N], s_init = sel[2 * N], s_last = sel[3 * N];
    gl_t tot[2];
    for (int c = 0; c < 2; c++) {
        const gl_t* ap = p.alpha_pows + (size_t)c * GLQ_MAX_TERMS;
        const gl_t ca = p.deltas[4 * c], cb = p.deltas[4 * c + 1], calpha = p.deltas[4 * c + 2], cdelta = p.deltas[4 * c + 3];
        gl_t acc = 0;
        for (int ch = 0; ch < 2; ch++) {
            // terms of challenge `ch` weighted by THIS output's alpha: t = 22 + (16 + num_luts) ch + k
            const gl_t da = p.deltas[4 * ch], db = p.deltas[4 * ch + 1], dalpha = p.deltas[4 * ch + 2], ddelta = p.deltas[4 * ch + 3];
            const gl_t* lz = p.zs + i + (size_t)(2 * GLP_CHUNKS + ch * NLP) * N;
            const gl_t* lzn = p.zs + i_next + (size_t)(2 * GLP_CHUNKS + ch * NLP) * N;
            const gl_t z_re = lz[0], next_z_re = lzn[0];
            uint32_t t = 2 + 2 * GLP_CHUNKS + GLQ_LOOKUP_TERMS(p.num_luts) * ch;
            auto term = [&](gl_t v) { acc = gl_add(acc, gl_mul(v, ap[t])); t++; };
            term(gl_mul(s_last, lz[(size_t)num_sldc * N]));                       // last LDC: z_x_sldc[num_sldc - 1]
            term(gl_mul(s_init, lz[N]));                                          // initial Sum: z_x_sldc[0]
            term(gl_mul(s_init, z_re));                                           // initial RE
            for (uint32_t tb = 0; tb < p.num_luts; tb++)                          // final RE: one per table, on the table's end selector
                term(gl_mul(sel[(size_t)(4 + tb) * N], gl_sub(z_re, p.lut_poly_at_delta[ch][tb])));
            gl_t cur = next_z_re;                                                 // RE transition
#pragma unroll 1
            for (int sl = 0; sl < 26; sl++) cur = gl_add(gl_mul(cur, ddelta), gl_add(w[(size_t)(3 * sl) * N], gl_mul(db, w[(size_t)(3 * sl + 1) * N])));
            term(gl_mul(s_trans_sre, gl_sub(z_re, cur)));
#pragma unroll 1
            for (uint32_t poly = 0; poly < num_sldc; poly++) {
                // alpha - (inp + a out) of the table slots [5 poly, min(5 poly + 5, 26)) and of the lookup slots [7 poly, min(7 poly + 7, 40))
                const uint32_t t0 = poly * 5, t1 = (t0 + 5 < 26) ? t0 + 5 : 26, u0 = poly * 7, u1 = (u0 + 7 < 40) ? u0 + 7 : 40;
                gl_t ft[5], fu[7];
                for (uint32_t a = t0; a < t1; a++) ft[a - t0] = gl_sub(dalpha, gl_add(w[(size_t)(3 * a) * N], gl_mul(da, w[(size_t)(3 * a + 1) * N])));
                for (uint32_t a = u0; a < u1; a++) fu[a - u0] = gl_sub(dalpha, gl_add(w[(size_t)(2 * a) * N], gl_mul(da, w[(size_t)(2 * a + 1) * N])));
                gl_t lut_prod = 1, lu_prod = 1, lu_sum = 0, lut_sum_mul = 0;
                for (uint32_t a = 0; a < t1 - t0; a++) lut_prod = gl_mul(lut_prod, ft[a]);
                for (uint32_t a = 0; a < u1 - u0; a++) lu_prod = gl_mul(lu_prod, fu[a]);
                for (uint32_t a = 0; a < u1 - u0; a++) { gl_t pr = 1; for (uint32_t b = 0; b < u1 - u0; b++) if (b != a) pr = gl_mul(pr, fu[b]); lu_sum = gl_add(lu_sum, pr); }
                for (uint32_t a = 0; a < t1 - t0; a++) { gl_t pr = 1; for (uint32_t b = 0; b < t1 - t0; b++) if (b != a) pr = gl_mul(pr, ft[b]); lut_sum_mul = gl_add(lut_sum_mul, gl_mul(w[(size_t)(3 * (t0 + a) + 2) * N], pr)); }
                const gl_t prev = poly == 0 ? lzn[(size_t)num_sldc * N] : lz[(size_t)poly * N];
                const gl_t dz = gl_sub(lz[(size_t)(poly + 1) * N], prev);
                term(gl_mul(s_trans_sre, gl_sub(gl_mul(lut_prod, dz), lut_sum_mul)));
                term(gl_mul(s_trans_ldc, gl_add(gl_mul(lu_prod, dz), lu_sum)));
            }
            (void)ca; (void)cb; (void)calpha; (void)cdelta;
        }
        tot[c] = acc;
    }
    const gl_t zi = p.zh_inv[i & 7];
    p.out[i] = gl_canon(gl_add(p.out[i], gl_mul(tot[0], zi)));
    p.out[N + i] = gl_canon(gl_add(p.out[N + i], gl_mul(tot[1], zi)));
}

// ---- RandomAccessGate (gates/random_access.rs:139-184) ------------------------------------------------------------------------------
// A launch of its own, made only for circuits that have the gate (like the PoseidonGate and lookup launches): its list folding would
// otherwise set the register budget of k_quotient<false> for every circuit.  Adds sum_g filter_g * sum_j alpha^(T0 + j) c_{g,j} / Z_H(x)
// of the RandomAccessGates to both outputs.  BITS = gate_params[g]; per copy: BITS boolean checks, the index reconstruction, the folded
// list against the claimed element; then the extra constants.
template <int BITS>
__device__ __forceinline__ void glq_random_access_gate(const gl_t* w, const gl_t* gc, size_t N, GlAlphaAcc& acc, uint32_t T0) {
    const glhost::RandomAccessLayout ra(BITS);
    constexpr int VEC = 1 << BITS;
    uint32_t t = T0;
#pragma unroll 1
    for (uint32_t copy = 0; copy < ra.num_copies; copy++) {
        gl_t b[BITS];
#pragma unroll
        for (int k = 0; k < BITS; k++) { b[k] = w[(size_t)ra.wire_bit(k, copy) * N]; acc.add(t++, glx_mul<true>(b[k], glx_sub_cc(b[k], 1))); }
        gl_t rec = 0;
#pragma unroll
        for (int k = BITS - 1; k >= 0; k--) rec = glx_add_cc(glx_add_cc(rec, rec), b[k]);
        acc.add(t++, glx_sub_cc(rec, w[(size_t)ra.wire_access_index(copy) * N]));
        // fold the list pairwise by bit 0, then bit 1, ...: x + b (y - x).  The first fold reads the 2^BITS items from memory.
        gl_t items[VEC / 2];
#pragma unroll
        for (int j = 0; j < VEC / 2; j++) {
            const gl_t x = w[(size_t)ra.wire_list_item(2 * j, copy) * N], y = w[(size_t)ra.wire_list_item(2 * j + 1, copy) * N];
            items[j] = glx_add_cc(x, glx_mul<true>(b[0], glx_sub_cc(y, x)));
        }
#pragma unroll
        for (int k = 1; k < BITS; k++) {
#pragma unroll
            for (int j = 0; j < (VEC >> (k + 1)); j++) items[j] = glx_add_cc(items[2 * j], glx_mul<true>(b[k], glx_sub_cc(items[2 * j + 1], items[2 * j])));
        }
        acc.add(t++, glx_sub_cc(items[0], w[(size_t)ra.wire_claimed_element(copy) * N]));
    }
    for (uint32_t k = 0; k < ra.num_extra_constants; k++) acc.add(t++, glx_sub_cc(gc[(size_t)k * N], w[(size_t)ra.wire_extra_constant(k) * N]));
}
__global__ __launch_bounds__(256) void k_quotient_random_access(GlQuotParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t N = size_t(1) << p.lgN;
    if (i >= N) return;
    const gl_t* w = p.wires + i;
    const gl_t* cs = p.cs + i;
    const gl_t* gc = cs + (size_t)(p.num_selectors + p.num_lookup_selectors) * N;
    gl_t tot0 = 0, tot1 = 0;
#pragma unroll 1
    for (uint32_t g = 0; g < p.num_gates; g++) {
        if (p.gate_types[g] != 9) continue;
        const gl_t sel = cs[(size_t)p.gate_sel[g] * N];
        gl_t filter = 1;                                            // gate.rs:277-284
        for (uint32_t k = p.group_start[g]; k < p.group_end[g]; k++) if (k != g) filter = glx_mul<true>(filter, glx_sub_cc((gl_t)k, sel));
        if (p.num_selectors > 1) filter = glx_mul<true>(filter, glx_sub_cc((gl_t)0xFFFFFFFFull, sel));
        GlAlphaAcc acc; acc.start(p.alpha_pows);
        switch (p.gate_params[g]) {
            case 1: glq_random_access_gate<1>(w, gc, N, acc, p.gate_term0); break;
            case 2: glq_random_access_gate<2>(w, gc, N, acc, p.gate_term0); break;
            case 3: glq_random_access_gate<3>(w, gc, N, acc, p.gate_term0); break;
            case 4: glq_random_access_gate<4>(w, gc, N, acc, p.gate_term0); break;
            case 5: glq_random_access_gate<5>(w, gc, N, acc, p.gate_term0); break;
            default: glq_random_access_gate<6>(w, gc, N, acc, p.gate_term0); break;
        }
        gl_t fs0, fs1, unused;
        glx_mul3<true>(filter, acc.sum(0), filter, acc.sum(1), 0, 0, fs0, fs1, unused);
        tot0 = glx_add_cc(tot0, fs0); tot1 = glx_add_cc(tot1, fs1);
    }
    const gl_t zi = p.zh_inv[i & 7];
    gl_t o0, o1, unused;
    glx_mul3<true>(tot0, zi, tot1, zi, 0, 0, o0, o1, unused);
    p.out[i] = glx_add_cc(p.out[i], o0);
    p.out[N + i] = glx_add_cc(p.out[N + i], o1);
}

// compute_lookup_polys (plonk/prover.rs:425-541).  Step 1, in parallel over the lookup rows: grid = (rows from last_lu_row to first_lut_row,
// 2 challenges), block = 64 (slot = thread).  Per row and challenge it leaves eight words agg[ch][row - last_lu_row][8]:
//   [0]     table rows: sum_s (inp_s + b out_s) delta^(25 - s), the row's contribution to RE (RE[row] = RE[row + 1] delta^26 + this)
//   [1..6]  the row's six partial sums: table rows sum_{s in group} mult_s / (alpha - (inp_s + a out_s)) over groups of 5 slots,
//           LookupGate rows sum_{s in group} 1 / (alpha - (inp_s + a out_s)) over groups of 7 slots
// so that the sequential step 2 reads 8 words per row instead of 78 wires (sums in F_p are exact in any order).
__global__ __launch_bounds__(64) void k_lookup_inverses(const gl_t* __restrict__ wires, uint32_t n, uint32_t last_lu_row, uint32_t last_lut_row,
                                                        gl_t a0, gl_t alpha0, gl_t b0, gl_t delta0, gl_t a1, gl_t alpha1, gl_t b1, gl_t delta1,
                                                        gl_t* __restrict__ agg) {
    __shared__ gl_t term[64], re_term[64];
    const uint32_t row = last_lu_row + blockIdx.x, ch = blockIdx.y, sl = threadIdx.x;
    const bool table_row = row >= last_lut_row;
    const uint32_t nslots = table_row ? 26u : 40u, stride = table_row ? 3u : 2u, group = table_row ? 5u : 7u;
    gl_t v = 0, rt = 0;
    if (sl < nslots) {
        const gl_t inp = wires[(size_t)(stride * sl) * n + row], outv = wires[(size_t)(stride * sl + 1) * n + row];
        const gl_t a = ch ? a1 : a0, alpha = ch ? alpha1 : alpha0;
        v = gl_inv(gl_sub(alpha, gl_add(inp, gl_mul(a, outv))));
        if (table_row) {
            v = gl_mul(v, wires[(size_t)(3 * sl + 2) * n + row]);                                      // multiplicity / (alpha - combo)
            rt = gl_mul(gl_add(inp, gl_mul(ch ? b1 : b0, outv)), gl_exp(ch ? delta1 : delta0, 25u - sl));
        }
    }
    term[sl] = v; re_term[sl] = rt;
    __syncthreads();
    gl_t* o = agg + ((size_t)ch * gridDim.x + blockIdx.x) * 8;
    if (sl == 0) { gl_t s = 0; for (uint32_t k = 0; k < 26; k++) s = gl_add(s, re_term[k]); o[0] = gl_canon(s); }
    if (sl >= 1 && sl <= 6) {
        const uint32_t s0 = (sl - 1) * group, s1 = (s0 + group < nslots) ? s0 + group : nslots;
        gl_t s = 0;
        for (uint32_t k = s0; k < s1; k++) s = gl_add(s, term[k]);
        o[sl] = gl_canon(s);
    }
}
// Step 2: the running sums, sequential over the (few) rows: thread c = challenge c, eight aggregate words per row.  Writes the 7
// columns of its challenge (RE, 6 partial SLDC), which the caller zeroed: out[(7 c + k) * n + row].
__global__ void k_lookup_scan(uint32_t n, uint32_t last_lu_row, uint32_t last_lut_row, uint32_t first_lut_row,
                              gl_t delta0, gl_t delta1, const gl_t* __restrict__ agg, gl_t* __restrict__ out) {
    const uint32_t c = threadIdx.x;
    if (c >= 2) return;
    const gl_t d26 = gl_exp(c ? delta1 : delta0, 26);
    const uint32_t nrows = first_lut_row - last_lu_row + 1;
    const gl_t* ag = agg + (size_t)c * nrows * 8;
    gl_t* o = out + (size_t)(7 * c) * n;
    gl_t re_next = 0, last_next = 0;                           // RE and the last partial polynomial on the row above (0 above the first table row)
    for (uint32_t row = first_lut_row + 1; row-- > last_lut_row;) {          // partial Sums and RE, from the first table row down
        const gl_t* r = ag + (size_t)(row - last_lu_row) * 8;
        const gl_t re = gl_add(gl_mul(re_next, d26), r[0]);
        o[row] = gl_canon(re); re_next = re;
        gl_t sum = last_next;
        for (uint32_t slot = 0; slot < 6; slot++) { sum = gl_add(sum, r[1 + slot]); o[(size_t)(slot + 1) * n + row] = gl_canon(sum); }
        last_next = sum;
    }
    for (uint32_t row = last_lut_row; row-- > last_lu_row;) {                // partial LDCs
        const gl_t* r = ag + (size_t)(row - last_lu_row) * 8;
        gl_t cur = last_next;
        for (uint32_t slot = 0; slot < 6; slot++) { cur = gl_sub(cur, r[1 + slot]); o[(size_t)(slot + 1) * n + row] = gl_canon(cur); }
        last_next = cur;
    }
}

// ---- openings: p(z) in F_p^2 for `npolys` base polynomials of n coefficients ---------------------------------------
// grid = (npolys); block = 256: thread t does Horner over its strip, strips are combined with z^(strip start)
__global__ __launch_bounds__(256) void k_eval_at_ext(const gl_t* coeffs, uint32_t n, uint64_t stride, gl_t za, gl_t zb, gl_t* out /* [npolys][2] */) {
    __shared__ gl_t sha[256], shb[256];
    const uint32_t poly = blockIdx.x, t = threadIdx.x;
    const gl_t* c = coeffs + (uint64_t)poly * stride;
    const gl2_t z = gl2_make(za, zb);
    const uint32_t strip = (n + 255) / 256;
    const uint32_t lo = t * strip, hi = (lo + strip < n) ? lo + strip : n;
    gl2_t acc = gl2_make(0, 0);
    for (uint32_t k = hi; k > lo; k--) {
        acc = gl2_mul(acc, z);
        acc.a = gl_add(acc.a, c[k - 1]);
    }
    if (lo < n) acc = gl2_mul(acc, gl2_exp(z, lo)); else acc = gl2_make(0, 0);
    sha[t] = acc.a; shb[t] = acc.b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < (uint32_t)s) { sha[t] = gl_add(sha[t], sha[t + s]); shb[t] = gl_add(shb[t], shb[t + s]); }
        __syncthreads();
    }
    if (t == 0) { out[2 * poly] = gl_canon(sha[0]); out[2 * poly + 1] = gl_canon(shb[0]); }
}

// table of powers of an extension element: out_a[i], out_b[i] = z^i, i < n
__global__ __launch_bounds__(256) void k_ext_powers(gl_t za, gl_t zb, uint32_t n, gl_t* out_a, gl_t* out_b) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gl2_t v = gl2_canon(gl2_exp(gl2_make(za, zb), i));
    out_a[i] = v.a; out_b[i] = v.b;
}
// two points in one launch (grid.y = 2): out = [a(z0) | b(z0) | a(z1) | b(z1)], n entries each
__global__ __launch_bounds__(256) void k_ext_powers2(gl_t z0a, gl_t z0b, gl_t z1a, gl_t z1b, uint32_t n, gl_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gl2_t z = blockIdx.y ? gl2_make(z1a, z1b) : gl2_make(z0a, z0b);
    const gl2_t v = gl2_canon(gl2_exp(z, i));
    gl_t* o = out + (size_t)blockIdx.y * 2 * n;
    o[i] = v.a; o[n + i] = v.b;
}
// p(z) = sum_i c_i z^i with the powers tabulated once per proof: two base-field multiply-adds per coefficient, coalesced
// reads (thread t takes i = t, t + 256, ...).  grid = (npolys); block = 256.
__global__ __launch_bounds__(256) void k_eval_with_powers(const gl_t* coeffs, uint32_t n, uint64_t stride, const gl_t* __restrict__ pa,
                                                          const gl_t* __restrict__ pb, gl_t* out /* [npolys][2] */) {
    __shared__ gl_t sha[256], shb[256];
    const uint32_t poly = blockIdx.x, t = threadIdx.x;
    const gl_t* c = coeffs + (uint64_t)poly * stride;
    // four independent accumulator pairs and all loads of a step issued before the arithmetic: the chain is latency-bound otherwise
    gl_t a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
    uint32_t i = t;
    for (; i + 768 < n; i += 1024) {
        const gl_t c0 = c[i], c1 = c[i + 256], c2 = c[i + 512], c3 = c[i + 768];
        const gl_t x0 = pa[i], x1 = pa[i + 256], x2 = pa[i + 512], x3 = pa[i + 768];
        const gl_t y0 = pb[i], y1 = pb[i + 256], y2 = pb[i + 512], y3 = pb[i + 768];
        a0 = gl_mul_add(a0, c0, x0); b0 = gl_mul_add(b0, c0, y0);
        a1 = gl_mul_add(a1, c1, x1); b1 = gl_mul_add(b1, c1, y1);
        a2 = gl_mul_add(a2, c2, x2); b2 = gl_mul_add(b2, c2, y2);
        a3 = gl_mul_add(a3, c3, x3); b3 = gl_mul_add(b3, c3, y3);
    }
    for (; i < n; i += 256) { const gl_t ci = c[i]; a0 = gl_mul_add(a0, ci, pa[i]); b0 = gl_mul_add(b0, ci, pb[i]); }
    sha[t] = gl_add(gl_add(a0, a1), gl_add(a2, a3)); shb[t] = gl_add(gl_add(b0, b1), gl_add(b2, b3));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < (uint32_t)s) { sha[t] = gl_add(sha[t], sha[t + s]); shb[t] = gl_add(shb[t], shb[t + s]); }
        __syncthreads();
    }
    if (t == 0) { out[2 * poly] = gl_canon(sha[0]); out[2 * poly + 1] = gl_canon(shb[0]); }
}

// the same for a LIST of polynomials in one launch: block j evaluates cols[j]; the first `n_first` at the point whose powers
// are (pa, pb), the rest at the point of (pa2, pb2) -- all openings of a proof (255 at zeta, 2 at g zeta) behind one launch
__global__ __launch_bounds__(256) void k_eval_list_with_powers(const gl_t* const* __restrict__ cols, uint32_t n, uint32_t n_first,
                                                               const gl_t* __restrict__ pa, const gl_t* __restrict__ pb,
                                                               const gl_t* __restrict__ pa2, const gl_t* __restrict__ pb2, gl_t* out /* [npolys][2] */) {
    __shared__ gl_t sha[256], shb[256];
    const uint32_t poly = blockIdx.x, t = threadIdx.x;
    const gl_t* c = cols[poly];
    if (poly >= n_first) { pa = pa2; pb = pb2; }
    gl_t a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0, b2 = 0, a3 = 0, b3 = 0;
    uint32_t i = t;
    for (; i + 768 < n; i += 1024) {
        const gl_t c0 = c[i], c1 = c[i + 256], c2 = c[i + 512], c3 = c[i + 768];
        const gl_t x0 = pa[i], x1 = pa[i + 256], x2 = pa[i + 512], x3 = pa[i + 768];
        const gl_t y0 = pb[i], y1 = pb[i + 256], y2 = pb[i + 512], y3 = pb[i + 768];
        a0 = gl_mul_add(a0, c0, x0); b0 = gl_mul_add(b0, c0, y0);
        a1 = gl_mul_add(a1, c1, x1); b1 = gl_mul_add(b1, c1, y1);
        a2 = gl_mul_add(a2, c2, x2); b2 = gl_mul_add(b2, c2, y2);
        a3 = gl_mul_add(a3, c3, x3); b3 = gl_mul_add(b3, c3, y3);
    }
    for (; i < n; i += 256) { const gl_t ci = c[i]; a0 = gl_mul_add(a0, ci, pa[i]); b0 = gl_mul_add(b0, ci, pb[i]); }
    sha[t] = gl_add(gl_add(a0, a1), gl_add(a2, a3)); shb[t] = gl_add(gl_add(b0, b1), gl_add(b2, b3));
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < (uint32_t)s) { sha[t] = gl_add(sha[t], sha[t + s]); shb[t] = gl_add(shb[t], shb[t + s]); }
        __syncthreads();
    }
    if (t == 0) { out[2 * poly] = gl_canon(sha[0]); out[2 * poly + 1] = gl_canon(shb[0]); }
}

// ---- FRI: F = sum_j alpha^j f_j over a list of coefficient columns --------------------------------------------------
// cols: device array of npolys pointers; apow: [npolys][2]; out planes a[n], b[n] (accumulate = add into existing)
// a workgroup covers 64 coefficient indices; its four waves each sum a quarter of the polynomials (a wave reads 64
// consecutive coefficients of one column: coalesced), then the quarters are added through LDS: four times the waves and a
// quarter of the dependent multiply-add chain of "one thread per coefficient" (109 -> ~35 us for 257 x 2^15)
__global__ __launch_bounds__(256) void k_fri_combine(const gl_t* const* cols, const gl_t* apow, uint32_t npolys, uint32_t n, gl_t* out_a, gl_t* out_b) {
    __shared__ gl_t sa[4][64], sb[4][64];
    const uint32_t ii = threadIdx.x & 63, g = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64 + ii;
    gl_t a = 0, b = 0;
    if (i < n) {
        const uint32_t per = (npolys + 3) / 4, j0 = g * per, j1 = (j0 + per < npolys) ? j0 + per : npolys;
        for (uint32_t j = j0; j < j1; j++) {
            const gl_t c = cols[j][i];
            a = gl_mul_add(a, c, apow[2 * j]);
            b = gl_mul_add(b, c, apow[2 * j + 1]);
        }
    }
    sa[g][ii] = a; sb[g][ii] = b;
    __syncthreads();
    if (g == 0 && i < n) {
        out_a[i] = gl_canon(gl_add(gl_add(sa[0][ii], sa[1][ii]), gl_add(sa[2][ii], sa[3][ii])));
        out_b[i] = gl_canon(gl_add(gl_add(sb[0][ii], sb[1][ii]), gl_add(sb[2][ii], sb[3][ii])));
    }
}

// (F(X) - F(z)) / (X - z) by segmented backward Horner: b_i = b_{i+1} z + c_i, quotient[i] = b_{i+1}, quotient[n-1] = 0.
// Segments of `seg` coefficients (at most 1024 segments).
// D1: head value of each segment assuming zero carry-in:  L_s = sum_{k in seg} c_k z^(k - start)
__global__ void k_div_linear_heads(const gl_t* ca, const gl_t* cb, uint32_t n, uint32_t seg, gl_t za, gl_t zb, gl_t* heads /* [nseg][2] */) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, nseg = (n + seg - 1) / seg;
    if (s >= nseg) return;
    const gl2_t z = gl2_make(za, zb);
    const uint32_t lo = s * seg, hi = (lo + seg < n) ? lo + seg : n;
    gl2_t acc = gl2_make(0, 0);
    for (uint32_t k = hi; k > lo; k--) acc = gl2_add(gl2_mul(acc, z), gl2_make(ca[k - 1], cb[k - 1]));
    heads[2 * s] = acc.a; heads[2 * s + 1] = acc.b;
}
// D2: carry-in of every segment.  With f_s(x) = L_s + z^len(s) x the value of b at the start of segment s is
// (f_s o f_{s+1} o ... o f_last)(0); compositions of affine maps are associative -> suffix scan in LDS (one workgroup).
__global__ __launch_bounds__(1024) void k_div_linear_carries(gl_t* heads, uint32_t n, uint32_t seg, gl_t za, gl_t zb) {
    __shared__ gl_t La[1024], Lb[1024], Za[1024], Zb[1024];
    const uint32_t s = threadIdx.x, nseg = (n + seg - 1) / seg;
    const gl2_t z = gl2_make(za, zb);
    gl2_t L = gl2_make(0, 0), Z = gl2_make(1, 0);
    if (s < nseg) {
        const uint32_t lo = s * seg, len = (lo + seg < n) ? seg : n - lo;
        L = gl2_make(heads[2 * s], heads[2 * s + 1]);
        Z = gl2_exp(z, len);
    }
    La[s] = L.a; Lb[s] = L.b; Za[s] = Z.a; Zb[s] = Z.b;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        gl2_t L2 = gl2_make(0, 0), Z2 = gl2_make(1, 0);
        const bool has = s + d < 1024;
        if (has) { L2 = gl2_make(La[s + d], Lb[s + d]); Z2 = gl2_make(Za[s + d], Zb[s + d]); }
        __syncthreads();
        if (has) {      // (L, Z) o (L2, Z2) = (L + Z L2, Z Z2)
            L = gl2_add(L, gl2_mul(Z, L2));
            Z = gl2_mul(Z, Z2);
            La[s] = L.a; Lb[s] = L.b; Za[s] = Z.a; Zb[s] = Z.b;
        }
        __syncthreads();
    }
    // carry-in of segment s = b at the start of segment s+1
    if (s < nseg) {
        gl_t ca = 0, cb = 0;
        if (s + 1 < nseg) { ca = La[s + 1]; cb = Lb[s + 1]; }
        heads[2 * s] = gl_canon(ca); heads[2 * s + 1] = gl_canon(cb);
    }
}
// D3: exact recurrence inside each segment, writing scale * quotient (+ what is already there if accumulate)
__global__ void k_div_linear_apply(const gl_t* ca, const gl_t* cb, uint32_t n, uint32_t seg, gl_t za, gl_t zb, const gl_t* carries,
                                   gl_t sa, gl_t sb, gl_t* qa, gl_t* qb, int accumulate) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, nseg = (n + seg - 1) / seg;
    if (s >= nseg) return;
    const gl2_t z = gl2_make(za, zb), scale = gl2_make(sa, sb);
    const uint32_t lo = s * seg, hi = (lo + seg < n) ? lo + seg : n;
    gl2_t b = gl2_make(carries[2 * s], carries[2 * s + 1]);          // b_hi
    for (uint32_t k = hi; k > lo; k--) {
        gl2_t q = gl2_mul(b, scale);                                  // quotient[k-1] = b_k
        if (accumulate) q = gl2_add(q, gl2_make(qa[k - 1], qb[k - 1]));
        qa[k - 1] = gl_canon(q.a); qb[k - 1] = gl_canon(q.b);
        b = gl2_add(gl2_mul(b, z), gl2_make(ca[k - 1], cb[k - 1]));
    }
}

// ---- FRI fold: out[k] = sum_{i < arity} beta^i in[arity k + i]   (fri/prover.rs:94-103, plonk_common.rs:116-128) ------
__global__ __launch_bounds__(256) void k_fri_fold(const gl_t* ia, const gl_t* ib, uint32_t n_out, uint32_t arity, gl_t ba, gl_t bb, gl_t* oa, gl_t* ob) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    const gl2_t beta = gl2_make(ba, bb);
    gl2_t acc = gl2_make(0, 0);
    for (uint32_t i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), gl2_make(ia[(size_t)arity * k + i], ib[(size_t)arity * k + i]));
    oa[k] = gl_canon(acc.a); ob[k] = gl_canon(acc.b);
}

// ---- proof of work: smallest w >= base with clz(permute(state with w at pos)[7]) >= bits ----------------------------------
struct GlPowParams { gl_t state[12]; uint32_t pos, min_leading_zeros; uint64_t base, count; unsigned long long* result; };
__global__ __launch_bounds__(256) void k_pow_grind(GlPowParams p) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.count) return;
    const uint64_t cand = p.base + idx;
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = p.state[i];
#pragma unroll
    for (int i = 0; i < 8; i++) if ((uint32_t)i == p.pos) s[i] = cand;
    psd_permute(s);
    const gl_t r = gl_canon(s[7]);
    const uint32_t lz = r ? (uint32_t)__clzll((long long)r) : 64u;
    if (lz >= p.min_leading_zeros) atomicMin(p.result, (unsigned long long)cand);
}

// ---- gathers for the query phase -----------------------------------------------------------------------------------------
// rows of a column-major matrix: out[q][c] = base[c * stride + rows[q]]
__global__ void k_gather_rows(const gl_t* base, uint64_t stride, uint32_t ncols, const uint32_t* rows, uint32_t nq, gl_t* out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nq * ncols) return;
    const uint32_t q = idx / ncols, c = idx % ncols;
    out[idx] = base[(uint64_t)c * stride + rows[q]];
}
// Merkle paths: out[q][l][4] = digests[level_off[l] + ((leaf[q] >> l) ^ 1)]
__global__ void k_gather_paths(const gl_t* digests, const uint64_t* level_off, uint32_t nlevels, const uint32_t* leaves, uint32_t nq, gl_t* out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nq * nlevels * 4) return;
    const uint32_t k = idx & 3, l = (idx >> 2) % nlevels, q = (idx >> 2) / nlevels;
    out[idx] = digests[4 * (level_off[l] + ((leaves[q] >> l) ^ 1u)) + k];
}
// FRI step leaves: leaf j of a tree over an ext SoA array = positions bitrev(arity*j + k): out[q][k][2]
__global__ void k_gather_fri_leaves(const gl_t* va, const gl_t* vb, uint32_t lg_len, uint32_t arity_bits, const uint32_t* leaves, uint32_t nq, gl_t* out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t arity = 1u << arity_bits;
    if (idx >= nq * arity) return;
    const uint32_t q = idx >> arity_bits, k = idx & (arity - 1);
    const uint32_t pos = (leaves[q] << arity_bits) | k;                       // index in the bit-reversed array
    const uint32_t nat = lg_len ? (__brev(pos) >> (32 - lg_len)) : 0;         // natural-order index
    out[2 * idx] = va[nat]; out[2 * idx + 1] = vb[nat];
}
