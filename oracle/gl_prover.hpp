// ORACLE (test infrastructure only).  CPU restatement of the reference prover, its Fiat-Shamir transcript,
// the native verifier and the proof wire format, for the matmul demo circuit (PoseidonGoldilocksConfig, D = 2).
// Follows:
//   plonky2/src/iop/challenger.rs:30-153, fri/challenges.rs:14-65                (Challenger)
//   plonky2/src/plonk/prover.rs:102-329 (prove), :332-416 (Z / partial products), :574-744 (quotient)
//   plonky2/src/plonk/vanishing_poly.rs:54-160,164-330, util/partial_products.rs:13-76, plonk_common.rs:61-128
//   field/src/zero_poly_coset.rs:19-60
//   plonky2/src/plonk/proof.rs:306-381 (OpeningSet), plonk/circuit_data.rs:440-600 (FRI instance)
//   plonky2/src/fri/oracle.rs:162-219 (prove_openings), util/reducing.rs:83-106, polynomial/division.rs:75-88
//   plonky2/src/fri/prover.rs:20-216 (commit phase, PoW, queries)
//   plonky2/src/plonk/verifier.rs:15-115, plonk/get_challenges.rs:26-87, fri/verifier.rs:21-260
//   plonky2/src/util/serialization/mod.rs:1202-1275,1332-1339,1367-1378,1409-1423,1443-1458,1477-1490,
//                                         1508-1519,1532-1546,1568-1581,1939-1981 (wire format)
#pragma once
#include "gl_circuit.hpp"

namespace orc {

// ReducingFactor::reduce_polys_base (util/reducing.rs:83-94): sum_j alpha^j * poly_j with FORWARD powers alpha^0, alpha^1, ...
static inline std::vector<Ext2> reduce_polys_base(Ext2 alpha, const std::vector<const std::vector<u64>*>& polys, size_t n) {
    std::vector<Ext2> comp(n, Ext2{0, 0});
    Ext2 ap = ext(1);
    for (auto* p : polys) { for (size_t i = 0; i < n && i < p->size(); i++) comp[i] = eadd(comp[i], escalar(ap, (*p)[i])); ap = emul(ap, alpha); }
    return comp;
}
// ReducingFactor::reduce (util/reducing.rs:36-42): Horner from the back, sum_j alpha^j * v_j
static inline Ext2 reduce_ext(Ext2 alpha, const std::vector<Ext2>& vals) {
    Ext2 acc{0, 0};
    for (size_t i = vals.size(); i-- > 0;) acc = eadd(emul(acc, alpha), vals[i]);
    return acc;
}
// PolynomialCoeffs::divide_by_linear (field/src/polynomial/division.rs:75-88): (p(X) - p(z)) / (X - z), len - 1 coefficients
static inline std::vector<Ext2> divide_by_linear(const std::vector<Ext2>& coeffs, Ext2 z) {
    std::vector<Ext2> bs(coeffs.size());
    Ext2 acc{0, 0};
    for (size_t i = coeffs.size(); i-- > 0;) { acc = eadd(emul(acc, z), coeffs[i]); bs[i] = acc; }
    return std::vector<Ext2>(bs.begin() + (bs.empty() ? 0 : 1), bs.end());
}


// ---- Challenger: duplex sponge, challenges pop from the END of the rate (challenger.rs:81-92,134-148) ----
struct Challenger {
    PState state; std::vector<u64> in, out;
    Challenger() { state.fill(0); }
    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        poseidon(state);
        out.assign(state.begin(), state.begin() + 8);
    }
    void observe(u64 x) { out.clear(); in.push_back(x); if (in.size() == 8) duplexing(); }
    void observe_ext(Ext2 x) { observe(x.a); observe(x.b); }
    void observe_digest(const Digest& d) { for (int i = 0; i < 4; i++) observe(d.e[i]); }
    void observe_cap(const std::vector<Digest>& cap) { for (auto& d : cap) observe_digest(d); }
    u64 challenge() { if (!in.empty() || out.empty()) duplexing(); u64 v = out.back(); out.pop_back(); return v; }
    Ext2 ext_challenge() { u64 a = challenge(), b = challenge(); return Ext2{a, b}; }
};

struct FriQueryStep { std::vector<Ext2> evals; std::vector<Digest> merkle_proof; };
struct FriQueryRound {
    std::vector<std::pair<std::vector<u64>, std::vector<Digest>>> initial;   // per oracle: (leaf, path)
    std::vector<FriQueryStep> steps;
};
struct FriProof {
    std::vector<std::vector<Digest>> commit_phase_merkle_caps;
    std::vector<FriQueryRound> query_round_proofs;
    std::vector<Ext2> final_poly;
    u64 pow_witness = 0;
};
struct OpeningSet {
    std::vector<Ext2> constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys;
    std::vector<Ext2> lookup_zs, lookup_zs_next;          // empty without lookups (proof.rs:300-301)
};
struct Proof {
    std::vector<Digest> wires_cap, zs_pp_cap, quotient_cap;
    OpeningSet openings;
    FriProof opening_proof;
    std::vector<u64> public_inputs;
};
// intermediates kept for parity tests against the HIP path
struct ProverTrace {
    std::vector<u64> betas, gammas, alphas, deltas; Ext2 zeta, fri_alpha; std::vector<Ext2> fri_betas;
    std::vector<std::vector<u64>> zs_partial_products;     // 20 columns of values
    std::vector<std::vector<u64>> quotient_chunks;         // 16 columns of coefficients
    std::vector<Ext2> final_poly_coeffs_initial;           // alpha^2 Q0 + Q1 before the LDE (n coefficients)
    std::vector<size_t> query_indices;
    Digest public_inputs_hash;
};

static inline Ext2 eval_poly_ext(const std::vector<u64>& coeffs, Ext2 z) {        // polynomial/mod.rs:157-162
    Ext2 acc{0, 0};
    for (size_t i = coeffs.size(); i-- > 0;) acc = eadd(emul(acc, z), Ext2{coeffs[i], 0});
    return acc;
}
static inline Ext2 eval_extpoly_ext(const std::vector<Ext2>& coeffs, Ext2 z) {
    Ext2 acc{0, 0};
    for (size_t i = coeffs.size(); i-- > 0;) acc = eadd(emul(acc, z), coeffs[i]);
    return acc;
}

// quotient_chunk_products / partial_products_and_z_gx (util/partial_products.rs:13-37)
static inline std::vector<u64> quotient_chunk_products(const std::vector<u64>& quotient_values, size_t max_degree) {
    std::vector<u64> out;
    for (size_t c = 0; c < quotient_values.size(); c += max_degree) {
        u64 p = 1;
        for (size_t j = c; j < std::min(quotient_values.size(), c + max_degree); j++) p = mul(p, quotient_values[j]);
        out.push_back(p);
    }
    return out;
}
static inline std::vector<u64> partial_products_and_z_gx(u64 z_x, const std::vector<u64>& chunk_products) {
    std::vector<u64> res;
    u64 acc = z_x;
    for (u64 c : chunk_products) { acc = mul(acc, c); res.push_back(acc); }
    return res;
}

// wires_permutation_partial_products_and_zs (prover.rs:359-416) for one challenge: returns 10 columns
// [pp_0 .. pp_8, Z]
static inline std::vector<std::vector<u64>> partial_products_and_z(const CircuitData& cd, const Witness& w, u64 beta, u64 gamma, unsigned threads) {
    const CommonData& cm = cd.common;
    const size_t n = cm.degree(), R = cm.config.num_routed_wires, deg = cm.quotient_degree_factor, chunks = (R + deg - 1) / deg;
    std::vector<std::vector<u64>> chunk_prod(n, std::vector<u64>(chunks));
    const size_t nc = cm.num_constants;
    parallel_for(n, threads, [&](size_t i) {
        u64 x = cd.subgroup[i];
        std::vector<u64> den(R), num(R);
        for (size_t j = 0; j < R; j++) {
            u64 wv = w.wire_values[j][i];
            num[j] = add(add(wv, mul(beta, mul(cm.k_is[j], x))), gamma);
            den[j] = add(add(wv, mul(beta, cd.constants_sigmas[nc + j][i])), gamma);
        }
        std::vector<u64> inv_den = batch_inverse(den), quotient_values(R);
        for (size_t j = 0; j < R; j++) quotient_values[j] = mul(num[j], inv_den[j]);
        chunk_prod[i] = quotient_chunk_products(quotient_values, deg);
    });
    std::vector<std::vector<u64>> cols(chunks, std::vector<u64>(n));
    u64 z_x = 1;
    for (size_t i = 0; i < n; i++) {
        std::vector<u64> pps_and_z_gx = partial_products_and_z_gx(z_x, chunk_prod[i]);
        for (size_t c = 0; c < chunks; c++) cols[c][i] = pps_and_z_gx[c];
        std::swap(z_x, cols[chunks - 1][i]);          // prover.rs:402-410: store Z(x), carry Z(gx)
    }
    return cols;
}

struct ZeroPolyOnCoset {                                 // zero_poly_coset.rs:19-60
    u64 n; size_t rate; std::vector<u64> evals, inverses;
    ZeroPolyOnCoset(unsigned n_log, unsigned rate_bits) {
        n = u64(1) << n_log; rate = size_t(1) << rate_bits;
        u64 g_pow_n = exp_pow2(GL_GENERATOR, n_log), w = primitive_root_of_unity(rate_bits), x = 1;
        for (size_t i = 0; i < rate; i++) { evals.push_back(sub(mul(g_pow_n, x), 1)); x = mul(x, w); }
        inverses = batch_inverse(evals);
    }
    u64 eval_inverse(size_t i) const { return inverses[i % rate]; }
    u64 eval_l_0(size_t i, u64 x) const { return mul(evals[i % rate], inv(mul(n, sub(x, 1)))); }
};

// check_partial_products (partial_products.rs:52-76), generic over the field
template <class K>
static inline void check_partial_products(const K* num, const K* den, size_t R, const K* partials, size_t num_prods,
                                          K z_x, K z_gx, size_t max_degree, std::vector<K>& out) {
    size_t chunks = (R + max_degree - 1) / max_degree;
    for (size_t c = 0; c < chunks; c++) {
        K prev = c == 0 ? z_x : partials[c - 1];
        K next = c == chunks - 1 ? z_gx : partials[c];
        K np = kconst<K>(1), dp = kconst<K>(1);
        for (size_t j = c * max_degree; j < std::min(R, (c + 1) * max_degree); j++) { np = kmul(np, num[j]); dp = kmul(dp, den[j]); }
        out.push_back(ksub(kmul(prev, np), kmul(next, dp)));
    }
    (void)num_prods;
}

// ---- lookup argument (log-derivative, Tip5 layout) -------------------------------------------------------------------
// get_lut_poly(...).eval(delta) (vanishing_poly.rs:31-49): the table's combos (inp + b * out), zero-padded to `degree` entries and
// REVERSED, as coefficients of a polynomial evaluated at the delta challenge: sum_i combo_i * delta^(degree - 1 - i)
static inline u64 lut_poly_eval(const std::vector<std::pair<uint16_t, uint16_t>>& lut, u64 b, u64 delta, size_t degree) {
    u64 acc = 0;                                         // Horner from the highest coefficient = combo_0
    for (size_t i = 0; i < degree; i++) acc = add(mul(acc, delta), i < lut.size() ? add(lut[i].first, mul(b, lut[i].second)) : 0);
    return acc;
}
// check_lookup_constraints (vanishing_poly.rs:337-500 / 503-670), generic over the field: 4 + #tables + 2 * #SLDC constraints
template <class K>
static inline void check_lookup_constraints(const CommonData& cm, const K* wires, const K* local_lookup_zs, const K* next_lookup_zs,
                                            const K* lookup_selectors, const u64* deltas /* 4 */, std::vector<K>& out) {
    const size_t lu_degree = cm.quotient_degree_factor - 1, num_sldc = cm.num_lookup_polys - 1;
    const size_t lut_degree = (LOOKUP_TABLE_SLOTS + num_sldc - 1) / num_sldc;
    const K z_re = local_lookup_zs[0], next_z_re = next_lookup_zs[0];
    const K* z_x = local_lookup_zs + 1; const K* z_gx = next_lookup_zs + 1;
    const u64 ca = deltas[LU_CH_A], cb = deltas[LU_CH_B], calpha = deltas[LU_CH_ALPHA], cdelta = deltas[LU_CH_DELTA];
    std::vector<K> looked(LOOKUP_TABLE_SLOTS), looking(LOOKUP_SLOTS), lookup_combo(LOOKUP_TABLE_SLOTS);
    for (size_t sl = 0; sl < LOOKUP_TABLE_SLOTS; sl++) {
        looked[sl] = kadd(wires[3 * sl], kscal(wires[3 * sl + 1], ca));
        lookup_combo[sl] = kadd(wires[3 * sl], kscal(wires[3 * sl + 1], cb));
    }
    for (size_t sl = 0; sl < LOOKUP_SLOTS; sl++) looking[sl] = kadd(wires[2 * sl], kscal(wires[2 * sl + 1], ca));
    out.push_back(kmul(lookup_selectors[LU_SEL_LAST_LDC], z_x[num_sldc - 1]));          // last LDC
    out.push_back(kmul(lookup_selectors[LU_SEL_INIT_SRE], z_x[0]));                     // initial Sum
    out.push_back(kmul(lookup_selectors[LU_SEL_INIT_SRE], z_re));                       // initial RE
    for (size_t r = LU_SEL_START_END; r < cm.num_lookup_selectors; r++) {               // final RE, one per table
        auto& lut = cm.luts[r - LU_SEL_START_END];
        const size_t rows = (lut.size() + LOOKUP_TABLE_SLOTS - 1) / LOOKUP_TABLE_SLOTS;
        const u64 f = lut_poly_eval(lut, cb, cdelta, LOOKUP_TABLE_SLOTS * rows);
        out.push_back(kmul(lookup_selectors[r], ksub(z_re, kconst<K>(f))));
    }
    K cur = next_z_re;                                                                   // RE row transition
    for (size_t sl = 0; sl < LOOKUP_TABLE_SLOTS; sl++) cur = kadd(kscal(cur, cdelta), lookup_combo[sl]);
    out.push_back(kmul(lookup_selectors[LU_SEL_TRANS_SRE], ksub(z_re, cur)));
    const K alpha = kconst<K>(calpha);
    for (size_t poly = 0; poly < num_sldc; poly++) {
        const size_t t0 = poly * lut_degree, t1 = std::min((poly + 1) * lut_degree, LOOKUP_TABLE_SLOTS);
        const size_t u0 = poly * lu_degree, u1 = std::min((poly + 1) * lu_degree, LOOKUP_SLOTS);
        K lut_prod = kconst<K>(1), lu_prod = kconst<K>(1);
        for (size_t i = t0; i < t1; i++) lut_prod = kmul(lut_prod, ksub(alpha, looked[i]));
        for (size_t i = u0; i < u1; i++) lu_prod = kmul(lu_prod, ksub(alpha, looking[i]));
        K lu_sum_prods = kconst<K>(0), lut_sum_prods_mul = kconst<K>(0);
        for (size_t i = u0; i < u1; i++) {
            K pr = kconst<K>(1);
            for (size_t j = u0; j < u1; j++) if (j != i) pr = kmul(pr, ksub(alpha, looking[j]));
            lu_sum_prods = kadd(lu_sum_prods, pr);
        }
        for (size_t i = t0; i < t1; i++) {
            K pr = kconst<K>(1);
            for (size_t j = t0; j < t1; j++) if (j != i) pr = kmul(pr, ksub(alpha, looked[j]));
            lut_sum_prods_mul = kadd(lut_sum_prods_mul, kmul(wires[3 * i + 2], pr));
        }
        const K prev = poly == 0 ? z_gx[num_sldc - 1] : z_x[poly - 1];
        out.push_back(kmul(lookup_selectors[LU_SEL_TRANS_SRE], ksub(kmul(lut_prod, ksub(z_x[poly], prev)), lut_sum_prods_mul)));
        out.push_back(kmul(lookup_selectors[LU_SEL_TRANS_LDC], kadd(kmul(lu_prod, ksub(z_x[poly], prev)), lu_sum_prods)));
    }
}

// compute_lookup_polys (prover.rs:425-541) for one challenge: RE and the partial SLDC polynomials as value columns
static inline std::vector<std::vector<u64>> compute_lookup_polys(const CircuitData& cd, const Witness& w, const u64* deltas /* 4 */) {
    const CommonData& cm = cd.common;
    const size_t degree = cm.degree(), max_lookup_degree = cm.config.max_quotient_degree_factor - 1;
    const size_t num_partial = (LOOKUP_SLOTS + max_lookup_degree - 1) / max_lookup_degree;
    const size_t max_table_degree = (LOOKUP_TABLE_SLOTS + num_partial - 1) / num_partial;
    std::vector<std::vector<u64>> polys(num_partial + 1, std::vector<u64>(degree, 0));
    auto wire = [&](size_t row, size_t col) { return w.wire_values[col][row]; };
    for (auto& lr : cd.lookup_rows) {
        for (size_t row = lr.first_lut_gate + 1; row-- > lr.last_lut_gate;) {           // partial Sums and RE, from the first LUT row down
            std::vector<u64> minus(LOOKUP_TABLE_SLOTS);
            for (size_t sl = 0; sl < LOOKUP_TABLE_SLOTS; sl++)
                minus[sl] = sub(deltas[LU_CH_ALPHA], add(wire(row, 3 * sl), mul(deltas[LU_CH_A], wire(row, 3 * sl + 1))));
            std::vector<u64> inv_c = batch_inverse(minus);
            u64 new_re = polys[0][row + 1];
            for (size_t sl = 0; sl < LOOKUP_TABLE_SLOTS; sl++)
                new_re = add(mul(new_re, deltas[LU_CH_DELTA]), add(wire(row, 3 * sl), mul(deltas[LU_CH_B], wire(row, 3 * sl + 1))));
            polys[0][row] = new_re;
            for (size_t slot = 0; slot < num_partial; slot++) {
                u64 sum = slot != 0 ? polys[slot][row] : polys[num_partial][row + 1];
                for (size_t sl = slot * max_table_degree; sl < std::min((slot + 1) * max_table_degree, LOOKUP_TABLE_SLOTS); sl++)
                    sum = add(sum, mul(wire(row, 3 * sl + 2), inv_c[sl]));
                polys[slot + 1][row] = sum;
            }
        }
        for (size_t row = lr.last_lut_gate; row-- > lr.last_lu_gate;) {                 // partial LDCs
            std::vector<u64> minus(LOOKUP_SLOTS);
            for (size_t sl = 0; sl < LOOKUP_SLOTS; sl++)
                minus[sl] = sub(deltas[LU_CH_ALPHA], add(wire(row, 2 * sl), mul(deltas[LU_CH_A], wire(row, 2 * sl + 1))));
            std::vector<u64> inv_c = batch_inverse(minus);
            for (size_t slot = 0; slot < num_partial; slot++) {
                const u64 prev = slot == 0 ? polys[num_partial][row + 1] : polys[slot][row];
                u64 sum = 0;
                for (size_t sl = slot * max_lookup_degree; sl < std::min((slot + 1) * max_lookup_degree, LOOKUP_SLOTS); sl++) sum = add(sum, inv_c[sl]);
                polys[slot + 1][row] = sub(prev, sum);
            }
        }
    }
    return polys;
}

// Values of the vanishing combination at one point, for each alpha (vanishing_poly.rs:54-160 / 164-330)
template <class K>
static inline std::vector<K> eval_vanishing_poly(const CommonData& cm, K x, K l_0_x, const K* local_constants, const K* wires,
                                                 const u64* pi_hash, const K* local_zs, const K* next_zs, const K* partial_products,
                                                 const K* s_sigmas, const std::vector<u64>& betas, const std::vector<u64>& gammas,
                                                 const std::vector<u64>& alphas, const K* local_lookup_zs = nullptr, const K* next_lookup_zs = nullptr,
                                                 const std::vector<u64>& deltas = std::vector<u64>()) {
    const size_t R = cm.config.num_routed_wires, nch = cm.config.num_challenges, np = cm.num_partial_products;
    std::vector<K> z1_terms, pp_terms, lookup_terms, constraint_terms(cm.num_gate_constraints);
    evaluate_gate_constraints<K>(cm.selectors, cm.num_gate_constraints, local_constants, wires, pi_hash, constraint_terms.data());
    std::vector<K> num(R), den(R);
    for (size_t i = 0; i < nch; i++) {
        z1_terms.push_back(kmul(l_0_x, ksub(local_zs[i], kconst<K>(1))));
        if (cm.num_lookup_polys)                             // vanishing_poly.rs:263-281
            check_lookup_constraints<K>(cm, wires, local_lookup_zs + i * cm.num_lookup_polys, next_lookup_zs + i * cm.num_lookup_polys,
                                        local_constants + cm.selectors.num_selectors(), deltas.data() + NUM_COINS_LOOKUP * i, lookup_terms);
        for (size_t j = 0; j < R; j++) {
            num[j] = kadd(kadd(wires[j], kscal(kscal(x, cm.k_is[j]), betas[i])), kconst<K>(gammas[i]));
            den[j] = kadd(kadd(wires[j], kscal(s_sigmas[j], betas[i])), kconst<K>(gammas[i]));
        }
        check_partial_products<K>(num.data(), den.data(), R, partial_products + i * np, np, local_zs[i], next_zs[i], cm.quotient_degree_factor, pp_terms);
    }
    std::vector<K> terms;
    terms.insert(terms.end(), z1_terms.begin(), z1_terms.end());
    terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
    terms.insert(terms.end(), lookup_terms.begin(), lookup_terms.end());
    terms.insert(terms.end(), constraint_terms.begin(), constraint_terms.end());
    std::vector<K> res(alphas.size(), kconst<K>(0));                      // reduce_with_powers_multi (plonk_common.rs:97-114)
    for (size_t t = terms.size(); t-- > 0;)
        for (size_t a = 0; a < alphas.size(); a++) res[a] = kadd(terms[t], kscal(res[a], alphas[a]));
    return res;
}

// compute_quotient_polys (prover.rs:576-744) + split into chunks (:245-258): 16 coefficient vectors of length n
static inline bool compute_quotient_chunks(const CircuitData& cd, const PolynomialBatch& wires_c, const PolynomialBatch& zs_c,
                                           const Digest& pi_hash, const std::vector<u64>& betas, const std::vector<u64>& gammas,
                                           const std::vector<u64>& alphas, unsigned threads, std::vector<std::vector<u64>>& out,
                                           const std::vector<u64>& deltas = std::vector<u64>()) {
    const CommonData& cm = cd.common;
    const unsigned qbits = log2_ceil(cm.quotient_degree_factor);
    assert(qbits == cm.config.rate_bits && "restatement covers step = 1 only");
    const size_t lde = size_t(1) << (cm.degree_bits + qbits), next_step = size_t(1) << qbits, nch = cm.config.num_challenges;
    const size_t nc = cm.num_constants, R = cm.config.num_routed_wires, np = cm.num_partial_products;
    ZeroPolyOnCoset zh(cm.degree_bits, qbits);
    std::vector<u64> points(lde);
    { u64 g = primitive_root_of_unity(cm.degree_bits + qbits), x = 1; for (size_t i = 0; i < lde; i++) { points[i] = x; x = mul(x, g); } }
    u64 pih[4]; for (int k = 0; k < 4; k++) pih[k] = canon(pi_hash.e[k]);
    std::vector<std::vector<u64>> qvals(nch, std::vector<u64>(lde));
    parallel_for(threads > 1 ? threads * 16 : 1, threads, [&](size_t part) {
        size_t parts = threads > 1 ? threads * 16 : 1;
        for (size_t i = part * lde / parts; i < (part + 1) * lde / parts; i++) {
            u64 x = mul(GL_GENERATOR, points[i]);
            const u64* cs = cd.constants_sigmas_commitment.get_lde_values(i, 1);
            const u64* wr = wires_c.get_lde_values(i, 1);
            const u64* zl = zs_c.get_lde_values(i, 1);
            const u64* zn = zs_c.get_lde_values((i + next_step) % lde, 1);
            u64 l0 = zh.eval_l_0(i, x);
            const size_t lk = nch * (1 + np);                     // lookup polynomials sit behind Z and the partial products (circuit_data.rs:450-459)
            std::vector<u64> r = eval_vanishing_poly<u64>(cm, x, l0, cs, wr, pih, zl, zn, zl + nch, cs + nc, betas, gammas, alphas, zl + lk, zn + lk, deltas);
            u64 di = zh.eval_inverse(i);
            for (size_t a = 0; a < nch; a++) qvals[a][i] = mul(r[a], di);
        }
    });
    (void)R; (void)np;
    out.clear();
    const size_t n = cm.degree();
    for (size_t a = 0; a < nch; a++) {
        coset_ifft_inplace(qvals[a], GL_GENERATOR);
        for (size_t i = cm.quotient_degree_factor * n; i < lde; i++) if (canon(qvals[a][i]) != 0) return false;   // trim_to_len
        for (size_t c = 0; c < cm.quotient_degree_factor; c++) out.emplace_back(qvals[a].begin() + c * n, qvals[a].begin() + (c + 1) * n);
    }
    return true;
}

static inline std::vector<Digest> cap_of(const PolynomialBatch& b) { return b.tree.cap(); }

// FRI leaf tree over extension values: leaf j = values[bitrev order][arity*j .. arity*(j+1)) flattened (fri/prover.rs:80-89)
static inline MerkleTree fri_tree(const std::vector<Ext2>& values_bitrev, size_t arity, unsigned cap_height) {
    size_t leaves = values_bitrev.size() / arity;
    std::vector<u64> flat(values_bitrev.size() * 2);
    for (size_t i = 0; i < values_bitrev.size(); i++) { flat[2 * i] = values_bitrev[i].a; flat[2 * i + 1] = values_bitrev[i].b; }
    return merkle_build(std::move(flat), leaves, arity * 2, cap_height);
}

static inline u64 grind_pow(const Challenger& ch, unsigned min_leading_zeros) {        // fri/prover.rs:115-160
    PState base = ch.state;
    for (size_t i = 0; i < ch.in.size(); i++) base[i] = ch.in[i];
    const size_t pos = ch.in.size();
    for (u64 cand = 0;; cand++) {                      // ascending scan = minimum witness (1-thread reference order)
        PState s = base; s[pos] = cand;
        poseidon(s);
        u64 resp = canon(s[7]);
        unsigned lz = resp == 0 ? 64 : __builtin_clzll(resp);
        if (lz >= min_leading_zeros) return cand;
    }
}

// prove() (prover.rs:102-329) from the full witness matrix
static inline bool prove(const CircuitData& cd, const Witness& w, unsigned threads, Proof& proof, ProverTrace* trace = nullptr) {
    const CommonData& cm = cd.common; const CircuitConfig& cfg = cm.config;
    const size_t n = cm.degree(), nch = cfg.num_challenges;
    Digest pi_hash = hash_no_pad(w.public_inputs.data(), w.public_inputs.size());
    PolynomialBatch wires_c = batch_from_values(w.wire_values, cfg.rate_bits, cfg.cap_height, threads);
    Challenger ch;
    ch.observe_digest(cd.circuit_digest);
    ch.observe_digest(pi_hash);
    ch.observe_cap(cap_of(wires_c));
    std::vector<u64> betas, gammas, alphas;
    for (size_t i = 0; i < nch; i++) betas.push_back(ch.challenge());
    for (size_t i = 0; i < nch; i++) gammas.push_back(ch.challenge());
    // lookups: 4 coins per challenge, betas and gammas reused for the first ones (prover.rs:166-184): [betas | gammas | additional]
    const bool has_lookup = !cm.luts.empty();
    std::vector<u64> deltas;
    if (has_lookup) {
        deltas = betas; deltas.insert(deltas.end(), gammas.begin(), gammas.end());
        for (size_t i = 0; i < NUM_COINS_LOOKUP * nch - 2 * nch; i++) deltas.push_back(ch.challenge());
    }
    // Z first, then all partial products (prover.rs:189-200)
    std::vector<std::vector<u64>> zs, pps;
    for (size_t i = 0; i < nch; i++) {
        auto cols = partial_products_and_z(cd, w, betas[i], gammas[i], threads);
        zs.push_back(cols.back()); cols.pop_back();
        for (auto& c : cols) pps.push_back(c);
    }
    std::vector<std::vector<u64>> zs_pp = zs;
    zs_pp.insert(zs_pp.end(), pps.begin(), pps.end());
    if (has_lookup)                                          // compute_all_lookup_polys (prover.rs:543-572), appended (:206-211)
        for (size_t i = 0; i < nch; i++) for (auto& col : compute_lookup_polys(cd, w, deltas.data() + NUM_COINS_LOOKUP * i)) zs_pp.push_back(col);
    PolynomialBatch zs_c = batch_from_values(zs_pp, cfg.rate_bits, cfg.cap_height, threads);
    ch.observe_cap(cap_of(zs_c));
    for (size_t i = 0; i < nch; i++) alphas.push_back(ch.challenge());
    std::vector<std::vector<u64>> qchunks;
    if (!compute_quotient_chunks(cd, wires_c, zs_c, pi_hash, betas, gammas, alphas, threads, qchunks, deltas)) return false;
    PolynomialBatch quot_c = batch_from_coeffs(qchunks, cfg.rate_bits, cfg.cap_height, threads);
    ch.observe_cap(cap_of(quot_c));
    Ext2 zeta = ch.ext_challenge();
    Ext2 g = Ext2{primitive_root_of_unity(cm.degree_bits), 0};
    if (eeq(eexp_pow2(zeta, cm.degree_bits), ext(1))) return false;             // prover.rs:280-283
    Ext2 gzeta = emul(g, zeta);
    // OpeningSet::new (proof.rs:306-344)
    const PolynomialBatch* oracles[4] = {&cd.constants_sigmas_commitment, &wires_c, &zs_c, &quot_c};
    auto eval_all = [&](const PolynomialBatch& b, Ext2 z) {
        std::vector<Ext2> r(b.ncols());
        parallel_for(b.ncols(), threads, [&](size_t c) { r[c] = eval_poly_ext(b.polynomials[c], z); });
        return r;
    };
    std::vector<Ext2> cs_eval = eval_all(*oracles[0], zeta), zs_eval = eval_all(zs_c, zeta), zs_next = eval_all(zs_c, gzeta);
    OpeningSet& os = proof.openings;
    os.constants.assign(cs_eval.begin(), cs_eval.begin() + cm.num_constants);
    os.plonk_sigmas.assign(cs_eval.begin() + cm.num_constants, cs_eval.end());
    os.wires = eval_all(wires_c, zeta);
    os.plonk_zs.assign(zs_eval.begin(), zs_eval.begin() + nch);
    os.plonk_zs_next.assign(zs_next.begin(), zs_next.begin() + nch);
    const size_t nzp = nch * (1 + cm.num_partial_products);      // Z's and partial products; lookup polynomials follow
    os.partial_products.assign(zs_eval.begin() + nch, zs_eval.begin() + nzp);
    os.lookup_zs.assign(zs_eval.begin() + nzp, zs_eval.end());
    os.lookup_zs_next.assign(zs_next.begin() + nzp, zs_next.end());
    os.quotient_polys = eval_all(quot_c, zeta);
    // observe_openings (fri/challenges.rs:15-22; proof.rs:345-381): zeta batch [.., quotient, lookup_zs], then [zs_next, lookup_zs_next]
    for (auto* v : {&os.constants, &os.plonk_sigmas, &os.wires, &os.plonk_zs, &os.partial_products, &os.quotient_polys, &os.lookup_zs}) for (auto& e : *v) ch.observe_ext(e);
    for (auto& e : os.plonk_zs_next) ch.observe_ext(e);
    for (auto& e : os.lookup_zs_next) ch.observe_ext(e);

    // prove_openings (fri/oracle.rs:162-219)
    Ext2 alpha = ch.ext_challenge();
    std::vector<Ext2> final_poly;                        // coefficient form
    auto reduce_and_divide = [&](const std::vector<const std::vector<u64>*>& polys, Ext2 point) {
        std::vector<Ext2> q = divide_by_linear(reduce_polys_base(alpha, polys, n), point);
        q.push_back(Ext2{0, 0});                          // oracle.rs:188-190: "+ 0" keeps the length a power of two
        return q;
    };
    {
        std::vector<const std::vector<u64>*> batch0, batch1;
        // fri_all_polys / fri_next_batch_polys (circuit_data.rs:564-597): lookup polynomials come LAST in both batches
        for (int o = 0; o < 4; o++) for (size_t c = 0; c < (o == 2 ? nzp : oracles[o]->polynomials.size()); c++) batch0.push_back(&oracles[o]->polynomials[c]);
        for (size_t c = nzp; c < zs_c.polynomials.size(); c++) batch0.push_back(&zs_c.polynomials[c]);
        for (size_t i = 0; i < nch; i++) batch1.push_back(&zs_c.polynomials[i]);
        for (size_t c = nzp; c < zs_c.polynomials.size(); c++) batch1.push_back(&zs_c.polynomials[c]);
        std::vector<Ext2> q0 = reduce_and_divide(batch0, zeta);
        final_poly = q0;                                  // shift of the empty polynomial is a no-op
        std::vector<Ext2> q1 = reduce_and_divide(batch1, gzeta);
        Ext2 shift = eexp_u64(alpha, batch1.size());      // reducing.rs:103-106
        for (size_t i = 0; i < n; i++) final_poly[i] = eadd(emul(final_poly[i], shift), q1[i]);
    }
    if (trace) trace->final_poly_coeffs_initial = final_poly;
    const size_t N = n << cfg.rate_bits;
    std::vector<Ext2> coeffs = final_poly; coeffs.resize(N, Ext2{0, 0});
    std::vector<Ext2> values = coeffs;
    ext_coset_fft_inplace(values, GL_GENERATOR);
    // fri_committed_trees (fri/prover.rs:69-112)
    FriProof& fp = proof.opening_proof;
    std::vector<MerkleTree> trees;
    std::vector<Ext2> fri_betas;
    u64 shift = GL_GENERATOR;
    for (unsigned arity_bits : cm.fri_reduction_arity_bits) {
        size_t arity = size_t(1) << arity_bits;
        reverse_index_bits_in_place(values);
        trees.push_back(fri_tree(values, arity, cfg.cap_height));
        ch.observe_cap(trees.back().cap());
        fp.commit_phase_merkle_caps.push_back(trees.back().cap());
        Ext2 beta = ch.ext_challenge();
        fri_betas.push_back(beta);
        std::vector<Ext2> folded(coeffs.size() / arity);
        for (size_t k = 0; k < folded.size(); k++) {      // reduce_with_powers (plonk_common.rs:116-128)
            Ext2 s{0, 0};
            for (size_t i = arity; i-- > 0;) s = eadd(emul(s, beta), coeffs[arity * k + i]);
            folded[k] = s;
        }
        coeffs = folded;
        shift = exp_u64(shift, arity);
        values = coeffs;
        ext_coset_fft_inplace(values, shift);
    }
    coeffs.resize(coeffs.size() >> cfg.rate_bits);
    for (auto& c : coeffs) ch.observe_ext(c);
    fp.final_poly = coeffs;
    fp.pow_witness = grind_pow(ch, cfg.proof_of_work_bits);
    ch.observe(fp.pow_witness);
    u64 pow_response = ch.challenge();
    (void)pow_response;
    // query rounds (fri/prover.rs:162-216)
    std::vector<size_t> idx;
    for (unsigned q = 0; q < cfg.num_query_rounds; q++) idx.push_back((size_t)(canon(ch.challenge()) % N));
    for (size_t x_index : idx) {
        FriQueryRound qr;
        for (int o = 0; o < 4; o++) {
            const MerkleTree& t = oracles[o]->tree;
            qr.initial.push_back({std::vector<u64>(t.leaf(x_index), t.leaf(x_index) + t.leaf_len), t.prove(x_index)});
        }
        size_t xi = x_index;
        for (size_t i = 0; i < trees.size(); i++) {
            unsigned ab = cm.fri_reduction_arity_bits[i];
            const MerkleTree& t = trees[i];
            FriQueryStep st;
            const u64* leaf = t.leaf(xi >> ab);
            for (size_t k = 0; k < t.leaf_len / 2; k++) st.evals.push_back(Ext2{leaf[2 * k], leaf[2 * k + 1]});
            st.merkle_proof = t.prove(xi >> ab);
            qr.steps.push_back(st);
            xi >>= ab;
        }
        fp.query_round_proofs.push_back(qr);
    }
    proof.wires_cap = cap_of(wires_c); proof.zs_pp_cap = cap_of(zs_c); proof.quotient_cap = cap_of(quot_c);
    proof.public_inputs = w.public_inputs;
    if (trace) {
        trace->betas = betas; trace->gammas = gammas; trace->alphas = alphas; trace->deltas = deltas; trace->zeta = zeta; trace->fri_alpha = alpha;
        trace->fri_betas = fri_betas; trace->zs_partial_products = zs_pp; trace->quotient_chunks = qchunks;
        trace->query_indices = idx; trace->public_inputs_hash = pi_hash;
    }
    return true;
}

// ---- wire format (util/serialization/mod.rs:1939-1981) -----------------------------------------------------
static inline void put_u64(std::vector<uint8_t>& o, u64 v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static inline void put_field(std::vector<uint8_t>& o, u64 v) { put_u64(o, canon(v)); }
static inline void put_ext_vec(std::vector<uint8_t>& o, const std::vector<Ext2>& v) { for (auto& e : v) { put_field(o, e.a); put_field(o, e.b); } }
static inline void put_digest(std::vector<uint8_t>& o, const Digest& d) { for (int i = 0; i < 4; i++) put_field(o, d.e[i]); }
static inline void put_cap(std::vector<uint8_t>& o, const std::vector<Digest>& c) { for (auto& d : c) put_digest(o, d); }
static inline void put_merkle_proof(std::vector<uint8_t>& o, const std::vector<Digest>& p) { o.push_back((uint8_t)p.size()); for (auto& d : p) put_digest(o, d); }
static inline std::vector<uint8_t> proof_to_bytes(const Proof& p) {
    std::vector<uint8_t> o;
    put_cap(o, p.wires_cap); put_cap(o, p.zs_pp_cap); put_cap(o, p.quotient_cap);
    const OpeningSet& os = p.openings;                                            // :1409-1423
    put_ext_vec(o, os.constants); put_ext_vec(o, os.plonk_sigmas); put_ext_vec(o, os.wires); put_ext_vec(o, os.plonk_zs);
    put_ext_vec(o, os.plonk_zs_next); put_ext_vec(o, os.lookup_zs); put_ext_vec(o, os.lookup_zs_next);      // lookup vectors before the partial products
    put_ext_vec(o, os.partial_products); put_ext_vec(o, os.quotient_polys);
    const FriProof& f = p.opening_proof;
    for (auto& c : f.commit_phase_merkle_caps) put_cap(o, c);
    for (auto& qr : f.query_round_proofs) {
        for (auto& lp : qr.initial) { for (u64 v : lp.first) put_field(o, v); put_merkle_proof(o, lp.second); }
        for (auto& st : qr.steps) { put_ext_vec(o, st.evals); put_merkle_proof(o, st.merkle_proof); }
    }
    put_ext_vec(o, f.final_poly);
    put_field(o, f.pow_witness);
    put_u64(o, p.public_inputs.size());
    for (u64 v : p.public_inputs) put_field(o, v);
    return o;
}

// ProofWithPublicInputs::from_bytes (plonk/proof.rs:112-123; util/serialization Read side): shapes come from CommonData
static inline bool proof_from_bytes(const CommonData& cm, const uint8_t* b, size_t len, Proof& p) {
    size_t pos = 0;
    auto get_u64 = [&](u64& v) { if (pos + 8 > len) return false; v = 0; for (int i = 0; i < 8; i++) v |= (u64)b[pos + i] << (8 * i); pos += 8; return true; };
    auto get_digests = [&](std::vector<Digest>& v, size_t n) { v.resize(n); for (auto& d : v) for (int k = 0; k < 4; k++) if (!get_u64(d.e[k])) return false; return true; };
    auto get_ext = [&](std::vector<Ext2>& v, size_t n) { v.resize(n); for (auto& e : v) if (!get_u64(e.a) || !get_u64(e.b)) return false; return true; };
    auto get_path = [&](std::vector<Digest>& v) { if (pos >= len) return false; size_t n = b[pos++]; return get_digests(v, n); };
    const CircuitConfig& cfg = cm.config;
    const size_t ncap = size_t(1) << cfg.cap_height, nch = cfg.num_challenges;
    if (!get_digests(p.wires_cap, ncap) || !get_digests(p.zs_pp_cap, ncap) || !get_digests(p.quotient_cap, ncap)) return false;
    OpeningSet& os = p.openings;
    if (!get_ext(os.constants, cm.num_constants) || !get_ext(os.plonk_sigmas, cfg.num_routed_wires) || !get_ext(os.wires, cfg.num_wires) ||
        !get_ext(os.plonk_zs, nch) || !get_ext(os.plonk_zs_next, nch) || !get_ext(os.lookup_zs, nch * cm.num_lookup_polys) ||
        !get_ext(os.lookup_zs_next, nch * cm.num_lookup_polys) || !get_ext(os.partial_products, nch * cm.num_partial_products) ||
        !get_ext(os.quotient_polys, nch * cm.quotient_degree_factor)) return false;
    FriProof& f = p.opening_proof;
    f.commit_phase_merkle_caps.resize(cm.fri_reduction_arity_bits.size());
    for (auto& c : f.commit_phase_merkle_caps) if (!get_digests(c, ncap)) return false;
    const size_t widths[4] = {cm.num_constants + cfg.num_routed_wires, cfg.num_wires, nch * (1 + cm.num_partial_products + cm.num_lookup_polys), nch * cm.quotient_degree_factor};
    f.query_round_proofs.resize(cfg.num_query_rounds);
    for (auto& qr : f.query_round_proofs) {
        qr.initial.resize(4);
        for (int o = 0; o < 4; o++) {
            qr.initial[o].first.resize(widths[o]);
            for (auto& v : qr.initial[o].first) if (!get_u64(v)) return false;
            if (!get_path(qr.initial[o].second)) return false;
        }
        qr.steps.resize(cm.fri_reduction_arity_bits.size());
        for (size_t i = 0; i < qr.steps.size(); i++) {
            if (!get_ext(qr.steps[i].evals, size_t(1) << cm.fri_reduction_arity_bits[i])) return false;
            if (!get_path(qr.steps[i].merkle_proof)) return false;
        }
    }
    if (!get_ext(f.final_poly, cm.final_poly_len()) || !get_u64(f.pow_witness)) return false;
    u64 npi = 0;
    if (!get_u64(npi) || npi != cm.num_public_inputs) return false;
    p.public_inputs.resize(npi);
    for (auto& v : p.public_inputs) if (!get_u64(v)) return false;
    return pos == len;
}

// ---- native verifier (plonk/verifier.rs:15-115, fri/verifier.rs:62-260) --------------------------------------
static inline const char* verify(const CommonData& cm, const std::vector<Digest>& constants_sigmas_cap, const Digest& circuit_digest, const Proof& p);
static inline const char* verify(const CircuitData& cd, const Proof& p) {
    return verify(cd.common, cd.constants_sigmas_commitment.tree.cap(), cd.circuit_digest, p);
}
// VerifierOnlyCircuitData = (constants_sigmas_cap, circuit_digest) (plonk/circuit_data.rs:333-340)
static inline const char* verify(const CommonData& cm, const std::vector<Digest>& constants_sigmas_cap, const Digest& circuit_digest, const Proof& p) {
    const CircuitConfig& cfg = cm.config;
    const size_t nch = cfg.num_challenges, n = cm.degree(), N = n << cfg.rate_bits;
    const OpeningSet& os = p.openings; const FriProof& fp = p.opening_proof;
    // shape (validate_shape.rs)
    if (p.wires_cap.size() != (size_t(1) << cfg.cap_height) || os.wires.size() != cfg.num_wires || os.constants.size() != cm.num_constants ||
        os.plonk_sigmas.size() != cfg.num_routed_wires || os.plonk_zs.size() != nch || os.plonk_zs_next.size() != nch ||
        os.partial_products.size() != nch * cm.num_partial_products || os.quotient_polys.size() != nch * cm.quotient_degree_factor ||
        os.lookup_zs.size() != nch * cm.num_lookup_polys || os.lookup_zs_next.size() != nch * cm.num_lookup_polys ||
        fp.commit_phase_merkle_caps.size() != cm.fri_reduction_arity_bits.size() || fp.query_round_proofs.size() != cfg.num_query_rounds ||
        fp.final_poly.size() != cm.final_poly_len() || p.public_inputs.size() != cm.num_public_inputs)
        return "malformed proof";
    Digest pi_hash = hash_no_pad(p.public_inputs.data(), p.public_inputs.size());
    // get_challenges (get_challenges.rs:26-87)
    Challenger ch;
    ch.observe_digest(circuit_digest); ch.observe_digest(pi_hash); ch.observe_cap(p.wires_cap);
    std::vector<u64> betas, gammas, alphas;
    for (size_t i = 0; i < nch; i++) betas.push_back(ch.challenge());
    for (size_t i = 0; i < nch; i++) gammas.push_back(ch.challenge());
    std::vector<u64> deltas;
    if (cm.num_lookup_polys) {
        deltas = betas; deltas.insert(deltas.end(), gammas.begin(), gammas.end());
        for (size_t i = 0; i < NUM_COINS_LOOKUP * nch - 2 * nch; i++) deltas.push_back(ch.challenge());
    }
    ch.observe_cap(p.zs_pp_cap);
    for (size_t i = 0; i < nch; i++) alphas.push_back(ch.challenge());
    ch.observe_cap(p.quotient_cap);
    Ext2 zeta = ch.ext_challenge();
    for (auto* v : {&os.constants, &os.plonk_sigmas, &os.wires, &os.plonk_zs, &os.partial_products, &os.quotient_polys, &os.lookup_zs}) for (auto& e : *v) ch.observe_ext(e);
    for (auto& e : os.plonk_zs_next) ch.observe_ext(e);
    for (auto& e : os.lookup_zs_next) ch.observe_ext(e);
    Ext2 fri_alpha = ch.ext_challenge();
    std::vector<Ext2> fri_betas;
    for (auto& cap : fp.commit_phase_merkle_caps) { ch.observe_cap(cap); fri_betas.push_back(ch.ext_challenge()); }
    for (auto& c : fp.final_poly) ch.observe_ext(c);
    ch.observe(fp.pow_witness);
    u64 pow_response = canon(ch.challenge());
    std::vector<size_t> idx;
    for (unsigned q = 0; q < cfg.num_query_rounds; q++) idx.push_back((size_t)(canon(ch.challenge()) % N));
    // vanishing(zeta) == Z_H(zeta) * t(zeta)  (verifier.rs:64-101)
    u64 pih[4]; for (int k = 0; k < 4; k++) pih[k] = canon(pi_hash.e[k]);
    Ext2 zeta_n = eexp_pow2(zeta, cm.degree_bits), z_h = esub(zeta_n, ext(1));
    Ext2 l0 = eeq(zeta, ext(1)) ? ext(1) : emul(z_h, einv(escalar(esub(zeta, ext(1)), (u64)n)));    // plonk_common.rs:61-71
    std::vector<Ext2> van = eval_vanishing_poly<Ext2>(cm, zeta, l0, os.constants.data(), os.wires.data(), pih, os.plonk_zs.data(),
                                                      os.plonk_zs_next.data(), os.partial_products.data(), os.plonk_sigmas.data(), betas, gammas, alphas,
                                                      os.lookup_zs.data(), os.lookup_zs_next.data(), deltas);
    for (size_t i = 0; i < nch; i++) {
        Ext2 t{0, 0};
        for (size_t k = cm.quotient_degree_factor; k-- > 0;) t = eadd(emul(t, zeta_n), os.quotient_polys[i * cm.quotient_degree_factor + k]);
        if (!eeq(van[i], emul(z_h, t))) return "vanishing polynomial identity fails at zeta";
    }
    // FRI
    unsigned lz = pow_response == 0 ? 64 : __builtin_clzll(pow_response);
    if (lz < cfg.proof_of_work_bits) return "invalid proof of work witness";
    Ext2 g = Ext2{primitive_root_of_unity(cm.degree_bits), 0}, gzeta = emul(g, zeta);
    // PrecomputedReducedOpenings (fri/verifier.rs:243-260)
    auto reduce = [&](const std::vector<Ext2>& vals) { Ext2 acc{0, 0}; for (size_t i = vals.size(); i-- > 0;) acc = eadd(emul(acc, fri_alpha), vals[i]); return acc; };
    std::vector<Ext2> batch0;
    for (auto* v : {&os.constants, &os.plonk_sigmas, &os.wires, &os.plonk_zs, &os.partial_products, &os.quotient_polys, &os.lookup_zs}) batch0.insert(batch0.end(), v->begin(), v->end());
    std::vector<Ext2> batch1v = os.plonk_zs_next;
    batch1v.insert(batch1v.end(), os.lookup_zs_next.begin(), os.lookup_zs_next.end());
    Ext2 red0 = reduce(batch0), red1 = reduce(batch1v);
    const std::vector<Digest>* caps[4] = {&constants_sigmas_cap, &p.wires_cap, &p.zs_pp_cap, &p.quotient_cap};
    const size_t nzp = nch * (1 + cm.num_partial_products);
    const size_t widths[4] = {cm.num_constants + cfg.num_routed_wires, cfg.num_wires, nzp + nch * cm.num_lookup_polys, nch * cm.quotient_degree_factor};
    const unsigned log_n = cm.degree_bits + cfg.rate_bits;
    for (unsigned q = 0; q < cfg.num_query_rounds; q++) {
        const FriQueryRound& qr = fp.query_round_proofs[q];
        size_t x_index = idx[q];
        if (qr.initial.size() != 4 || qr.steps.size() != cm.fri_reduction_arity_bits.size()) return "malformed query round";
        for (int o = 0; o < 4; o++) {
            if (qr.initial[o].first.size() != widths[o]) return "malformed initial leaf";
            if (!merkle_verify(qr.initial[o].first.data(), widths[o], x_index, *caps[o], qr.initial[o].second)) return "initial Merkle proof fails";
        }
        u64 subgroup_x = mul(GL_GENERATOR, exp_u64(primitive_root_of_unity(log_n), reverse_bits(x_index, log_n)));
        // fri_combine_initial (fri/verifier.rs:124-165)
        Ext2 sx{subgroup_x, 0};
        std::vector<Ext2> ev0, ev1;
        for (int o = 0; o < 4; o++) for (size_t c = 0; c < (o == 2 ? nzp : widths[o]); c++) ev0.push_back(Ext2{qr.initial[o].first[c], 0});
        for (size_t c = nzp; c < widths[2]; c++) ev0.push_back(Ext2{qr.initial[2].first[c], 0});      // lookup polynomials last
        for (size_t i = 0; i < nch; i++) ev1.push_back(Ext2{qr.initial[2].first[i], 0});
        for (size_t c = nzp; c < widths[2]; c++) ev1.push_back(Ext2{qr.initial[2].first[c], 0});
        Ext2 sum = emul(esub(reduce(ev0), red0), einv(esub(sx, zeta)));
        sum = emul(sum, eexp_u64(fri_alpha, ev1.size()));
        sum = eadd(sum, emul(esub(reduce(ev1), red1), einv(esub(sx, gzeta))));
        Ext2 old_eval = sum;
        for (size_t i = 0; i < cm.fri_reduction_arity_bits.size(); i++) {
            unsigned ab = cm.fri_reduction_arity_bits[i]; size_t arity = size_t(1) << ab;
            const std::vector<Ext2>& evals = qr.steps[i].evals;
            if (evals.size() != arity) return "malformed step";
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!eeq(evals[within], old_eval)) return "FRI consistency check fails";
            // compute_evaluation (fri/verifier.rs:21-47): interpolate {(coset_start*g^i, evals_rev[i])} at beta
            u64 gg = primitive_root_of_unity(ab);
            std::vector<Ext2> ev(evals);
            reverse_index_bits_in_place(ev);
            u64 coset_start = mul(subgroup_x, exp_u64(gg, arity - reverse_bits(within, ab)));
            std::vector<u64> xs(arity); { u64 y = 1; for (size_t k = 0; k < arity; k++) { xs[k] = mul(coset_start, y); y = mul(y, gg); } }
            Ext2 beta = fri_betas[i], acc{0, 0};
            for (size_t a = 0; a < arity; a++) {          // Lagrange form
                Ext2 numr = ext(1); u64 den = 1;
                for (size_t b2 = 0; b2 < arity; b2++) if (b2 != a) { numr = emul(numr, esub(beta, Ext2{xs[b2], 0})); den = mul(den, sub(xs[a], xs[b2])); }
                acc = eadd(acc, emul(ev[a], escalar(numr, inv(den))));
            }
            old_eval = acc;
            std::vector<u64> flat;
            for (auto& e : evals) { flat.push_back(e.a); flat.push_back(e.b); }
            if (!merkle_verify(flat.data(), flat.size(), coset_index, fp.commit_phase_merkle_caps[i], qr.steps[i].merkle_proof)) return "FRI step Merkle proof fails";
            subgroup_x = exp_pow2(subgroup_x, ab);
            x_index = coset_index;
        }
        if (!eeq(eval_extpoly_ext(fp.final_poly, Ext2{subgroup_x, 0}), old_eval)) return "final polynomial evaluation is invalid";
    }
    return nullptr;
}

}  // namespace orc
