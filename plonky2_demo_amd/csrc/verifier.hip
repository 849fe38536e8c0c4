// Native verifier for proofs of circuits over the demo's gate set (host code, as in the reference:
// plonky2/src/plonk/verifier.rs:15-115, plonk/get_challenges.rs:26-87, plonk/vanishing_poly.rs:54-160,
// plonk/validate_shape.rs, fri/verifier.rs:21-260, fri/validate_shape.rs, hash/merkle_proofs.rs:54-75).
//
// Input is exactly what the reference's `VerifierCircuitData::verify` sees: CommonCircuitData (here the gl_circuit_desc),
// VerifierOnlyCircuitData (constants_sigmas_cap, circuit_digest) and ProofWithPublicInputs::to_bytes().  The proof is
// checked in place: one pass turns the byte string into a flat table of canonical words plus offsets, and every later
// step indexes that table.  No GPU is involved (verification is milliseconds of sequential hashing).
#include "context.hpp"
#include "host_circuit.hpp"

namespace {

typedef gl2_t E;                                      // F_p[X]/(X^2 - 7)
inline E e_of(gl_t a) { return gl2_make(a, 0); }
inline E e_add(E x, E y) { return gl2_add(x, y); }
inline E e_sub(E x, E y) { return gl2_sub(x, y); }
inline E e_mul(E x, E y) { return gl2_mul(x, y); }
inline E e_scale(E x, gl_t s) { return gl2_scalar(x, s); }
inline bool e_eq(E x, E y) { x = gl2_canon(x); y = gl2_canon(y); return x.a == y.a && x.b == y.b; }
inline E e_sbox(E x) { E x2 = e_mul(x, x), x4 = e_mul(x2, x2); return e_mul(e_mul(x, x2), x4); }
inline E e_pow2k(E x, unsigned k) { for (unsigned i = 0; i < k; i++) x = e_mul(x, x); return x; }

// ---- transcript (iop/challenger.rs:30-153) ----
struct Transcript {
    gl_t sponge[12], pending[8], ready[8];
    int npending = 0, nready = 0;
    Transcript() { for (auto& s : sponge) s = 0; }
    void squeeze() {
        for (int i = 0; i < npending; i++) sponge[i] = pending[i];
        npending = 0;
        psd_permute(sponge);
        for (int i = 0; i < 8; i++) ready[i] = sponge[i];
        nready = 8;
    }
    void absorb(gl_t x) { nready = 0; pending[npending++] = x; if (npending == 8) squeeze(); }
    void absorb(const gl_t* v, size_t n) { for (size_t i = 0; i < n; i++) absorb(v[i]); }
    gl_t draw() { if (npending || !nready) squeeze(); return gl_canon(ready[--nready]); }
    E draw_ext() { E r; r.a = draw(); r.b = draw(); return r; }
};

// ---- Merkle path to a cap (hash/merkle_proofs.rs:54-75; leaf hash plonk/config.rs:55-66) ----
void hash_leaf(const gl_t* v, size_t n, gl_t out[4]) {
    if (n <= 4) { for (size_t i = 0; i < 4; i++) out[i] = i < n ? gl_canon(v[i]) : 0; return; }
    glhost::host_hash_no_pad(v, n, out);
}
bool path_opens_to_cap(const gl_t* leaf, size_t leaf_len, size_t index, const gl_t* siblings, size_t nsib, const gl_t* cap, size_t cap_len) {
    gl_t cur[4];
    hash_leaf(leaf, leaf_len, cur);
    for (size_t l = 0; l < nsib; l++) {
        const gl_t* sib = siblings + 4 * l;
        gl_t out[4];
        if (index & 1) psd_two_to_one(sib, cur, out); else psd_two_to_one(cur, sib, out);
        for (int k = 0; k < 4; k++) cur[k] = out[k];
        index >>= 1;
    }
    if (index >= cap_len) return false;
    for (int k = 0; k < 4; k++) if (cur[k] != gl_canon(cap[4 * index + k])) return false;
    return true;
}

// ---- the demo's gates over the extension field ----
// Poseidon layers on extension elements: every layer is F_p-linear except the S-box (hash/poseidon.rs:200-214,264-274,
// 311-366,429-450).  The partial rounds use this build's derived sparse factorisation; the constraint polynomials do not
// depend on the factorisation (only lane 0 meets the S-box, and lane 0 is the same in every factorisation).
void ext_mds(E (&s)[12]) {
    static const uint32_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    E out[12];
    for (int r = 0; r < 12; r++) {
        E acc = e_of(0);
        for (int i = 0; i < 12; i++) acc = e_add(acc, e_scale(s[(i + r) % 12], circ[i]));
        if (r == 0) acc = e_add(acc, e_scale(s[0], 8));
        out[r] = acc;
    }
    for (int i = 0; i < 12; i++) s[i] = out[i];
}
void ext_add_round_constants(E (&s)[12], int round) { for (int i = 0; i < 12; i++) s[i] = e_add(s[i], e_of(POSEIDON_RC[12 * round + i])); }

// 123 constraints of gates/poseidon.rs:113-191 on the row `w` (135 extension values); returns them in order
void poseidon_gate_constraints(const E* w, E* out) {
    using namespace glhost;
    int c = 0;
    const E swap = w[PW_SWAP];
    out[c++] = e_mul(swap, e_sub(swap, e_of(1)));
    for (int i = 0; i < 4; i++) out[c++] = e_sub(e_mul(swap, e_sub(w[PW_INPUT + 4 + i], w[PW_INPUT + i])), w[PW_DELTA + i]);
    E s[12];
    for (int i = 0; i < 4; i++) { s[i] = e_add(w[PW_INPUT + i], w[PW_DELTA + i]); s[4 + i] = e_sub(w[PW_INPUT + 4 + i], w[PW_DELTA + i]); s[8 + i] = w[PW_INPUT + 8 + i]; }
    int round = 0;
    for (int r = 0; r < 4; r++, round++) {
        ext_add_round_constants(s, round);
        if (r) for (int i = 0; i < 12; i++) { const E in = w[PW_FULL0 + 12 * (r - 1) + i]; out[c++] = e_sub(s[i], in); s[i] = in; }
        for (int i = 0; i < 12; i++) s[i] = e_sbox(s[i]);
        ext_mds(s);
    }
    for (int i = 0; i < 12; i++) s[i] = e_add(s[i], e_of(POSEIDON_PARTIAL_FIRST_RC[i]));
    {
        E t[12]; t[0] = s[0];
        for (int col = 1; col < 12; col++) {
            E acc = e_of(0);
            for (int r = 1; r < 12; r++) acc = e_add(acc, e_scale(s[r], POSEIDON_PARTIAL_INIT[(r - 1) * 11 + (col - 1)]));
            t[col] = acc;
        }
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        const E in = w[PW_PARTIAL + r];
        out[c++] = e_sub(s[0], in);
        const E s0 = e_add(e_sbox(in), e_of(POSEIDON_PARTIAL_RC[r]));
        E d = e_scale(s0, 25);
        for (int i = 1; i < 12; i++) d = e_add(d, e_scale(s[i], POSEIDON_PARTIAL_ROW[r * 11 + i - 1]));
        for (int i = 1; i < 12; i++) s[i] = e_add(s[i], e_scale(s0, POSEIDON_PARTIAL_COL[r * 11 + i - 1]));
        s[0] = d;
    }
    round += POSEIDON_PARTIAL_ROUNDS;
    for (int r = 0; r < 4; r++, round++) {
        ext_add_round_constants(s, round);
        for (int i = 0; i < 12; i++) { const E in = w[PW_FULL1 + 12 * r + i]; out[c++] = e_sub(s[i], in); s[i] = in; }
        for (int i = 0; i < 12; i++) s[i] = e_sbox(s[i]);
        ext_mds(s);
    }
    for (int i = 0; i < 12; i++) out[c++] = e_sub(s[i], w[PW_OUTPUT + i]);
}

struct Cursor {                                        // little-endian reader over the proof bytes
    const uint8_t* p; size_t len, pos = 0; bool ok = true;
    Cursor(const uint8_t* b, size_t n) : p(b), len(n) {}
    uint64_t u64() {
        if (pos + 8 > len) { ok = false; return 0; }
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)p[pos + i] << (8 * i);
        pos += 8;
        return v;
    }
    unsigned u8() { if (pos >= len) { ok = false; return 0; } return p[pos++]; }
};

int reject(const char* why) { return gl_fail(GL_ERR_VERIFY, why, __FILE__, __LINE__); }

}  // namespace

extern "C" int gl_verify(const gl_circuit_desc* desc, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4],
                         const uint8_t* proof_bytes, size_t num_bytes) {
    GL_REQUIRE(desc && constants_sigmas_cap && circuit_digest && proof_bytes, GL_ERR_ARG, "gl_verify: null argument");
    const gl_circuit_desc& d = *desc;
    GL_REQUIRE(d.num_wires == 135 && d.num_routed_wires == 80 && d.num_challenges == 2 && d.quotient_degree_factor == 8 && d.rate_bits == 3,
               GL_ERR_UNSUPPORTED, "gl_verify: only standard_recursion_config circuits are supported");
    GL_REQUIRE(d.num_gates >= 1 && d.num_gates <= GL_MAX_GATES, GL_ERR_ARG, "gl_verify: bad gate count");
    { const char* why = glhost::lookup_shape_error(d); GL_REQUIRE(!why, GL_ERR_UNSUPPORTED, why); }
    GL_REQUIRE(d.num_selectors >= 1 && d.num_constants == d.num_selectors + d.num_lookup_selectors + 2 && d.num_fri_rounds <= 8 &&
               d.degree_bits >= 1 && d.degree_bits + d.rate_bits <= 32 && d.cap_height <= d.degree_bits + d.rate_bits && d.num_query_rounds >= 1,
               GL_ERR_ARG, "gl_verify: bad circuit description");
    // every count that sizes an allocation below is bounded by what a proof of num_bytes can hold (a description is caller-filled, but a
    // wrong one must be refused, not turned into a 2^40-byte allocation: tools/sanitizer/data_fuzz.cpp)
    GL_REQUIRE(d.num_selectors <= GL_MAX_GATES && d.cap_height <= 16 && d.num_query_rounds <= num_bytes / 8 && d.num_public_inputs <= num_bytes / 8,
               GL_ERR_ARG, "gl_verify: a count of the description exceeds what the proof bytes can hold");
    for (unsigned g = 0; g < d.num_gates; g++)
        GL_REQUIRE(d.gate_types[g] <= glhost::G_LAST && d.gate_selector_index[g] < d.num_selectors && d.gate_group_start[g] <= g && g < d.gate_group_end[g] && d.gate_group_end[g] <= d.num_gates,
                   GL_ERR_ARG, "gl_verify: bad gate / selector description");
    const size_t nch = 2, R = 80, W = 135, QF = 8, NPP = 9;            // partial products per challenge: ceil(80 / 8) - 1
    const size_t ncap = size_t(1) << d.cap_height, ncs = d.num_constants + R;
    const unsigned lgn = d.degree_bits, lgN = lgn + d.rate_bits;
    const size_t n = size_t(1) << lgn, N = size_t(1) << lgN;
    unsigned total_arity = 0;
    for (unsigned r = 0; r < d.num_fri_rounds; r++) { GL_REQUIRE(d.fri_arity_bits[r] >= 1 && d.fri_arity_bits[r] <= 8, GL_ERR_ARG, "gl_verify: bad FRI arity"); total_arity += d.fri_arity_bits[r]; }
    GL_REQUIRE(total_arity <= lgn, GL_ERR_ARG, "gl_verify: FRI reduces below the final polynomial");
    const size_t final_len = size_t(1) << (lgn - total_arity);
    const size_t NLP = d.num_lookup_polys;                              // lookup polynomials per challenge, behind Z and the partial products
    const size_t nzp = nch * (1 + NPP);
    const size_t widths[4] = {ncs, W, nzp + nch * NLP, nch * QF};

    // ---- decode (util/serialization/mod.rs:1939-1981 read side, plonk/validate_shape.rs, fri/validate_shape.rs) ----
    // like the reference's read_field (from_canonical_u64 without a range check in release builds) a word >= p is taken mod p
    Cursor in(proof_bytes, num_bytes);
    std::vector<gl_t> T;                               // all words of the proof in wire order, canonical
    T.reserve(num_bytes / 8 + 8);
    auto words = [&](size_t k) { size_t at = T.size(); for (size_t i = 0; i < k && in.ok; i++) T.push_back(gl_canon(in.u64())); return at; };
    struct PathRef { size_t leaf, leaf_len, sib, nsib; };
    const size_t o_caps = words(3 * 4 * ncap);
    // OpeningSet in wire order (mod.rs:1409-1423): the lookup vectors sit between zs_next and the partial products
    const size_t o_const = words(2 * d.num_constants), o_sig = words(2 * R), o_wires = words(2 * W), o_zs = words(2 * nch), o_zsn = words(2 * nch),
                 o_lk = words(2 * nch * NLP), o_lkn = words(2 * nch * NLP), o_pp = words(2 * nch * NPP), o_quot = words(2 * nch * QF);
    const size_t o_fcaps = words((size_t)d.num_fri_rounds * 4 * ncap);
    std::vector<PathRef> init_paths((size_t)d.num_query_rounds * 4), step_paths((size_t)d.num_query_rounds * d.num_fri_rounds);
    for (unsigned q = 0; q < d.num_query_rounds && in.ok; q++) {
        for (int o = 0; o < 4; o++) {
            PathRef& pr = init_paths[q * 4 + o];
            pr.leaf = words(widths[o]); pr.leaf_len = widths[o];
            pr.nsib = in.u8(); pr.sib = words(4 * pr.nsib);
        }
        unsigned lg_cur = lgN;
        for (unsigned r = 0; r < d.num_fri_rounds; r++) {
            PathRef& pr = step_paths[q * d.num_fri_rounds + r];
            pr.leaf_len = size_t(2) << d.fri_arity_bits[r]; pr.leaf = words(pr.leaf_len);
            pr.nsib = in.u8(); pr.sib = words(4 * pr.nsib);
            lg_cur -= d.fri_arity_bits[r];
            if (in.ok && pr.nsib + d.cap_height != lg_cur) return reject("malformed proof: FRI step Merkle path has the wrong length");
        }
        for (int o = 0; o < 4 && in.ok; o++)
            if (init_paths[q * 4 + o].nsib + d.cap_height != lgN) return reject("malformed proof: initial Merkle path has the wrong length");
    }
    const size_t o_final = words(2 * final_len);
    const gl_t pow_witness = gl_canon(in.u64());
    const uint64_t npis = in.u64();
    if (!in.ok) return reject("malformed proof: truncated");
    if (npis != d.num_public_inputs) return reject("malformed proof: wrong number of public inputs");
    const size_t o_pis = words(npis);
    if (!in.ok || in.pos != num_bytes) return reject("malformed proof: length mismatch");
    auto ext_at = [&](size_t off, size_t i) { return gl2_make(T[off + 2 * i], T[off + 2 * i + 1]); };

    // ---- challenges (plonk/get_challenges.rs:26-87, fri/challenges.rs:24-64) ----
    gl_t pi_hash[4];
    glhost::host_hash_no_pad(T.data() + o_pis, npis, pi_hash);
    Transcript tr;
    tr.absorb(circuit_digest, 4); tr.absorb(pi_hash, 4); tr.absorb(&T[o_caps], 4 * ncap);
    gl_t betas[2], gammas[2], alphas[2];
    for (auto& b : betas) b = tr.draw();
    for (auto& g : gammas) g = tr.draw();
    // lookup coins (get_challenges.rs:51-63): [betas | gammas | 4 more], four per challenge
    gl_t deltas[8] = {betas[0], betas[1], gammas[0], gammas[1], 0, 0, 0, 0};
    if (NLP) for (int i = 4; i < 8; i++) deltas[i] = tr.draw();
    tr.absorb(&T[o_caps + 4 * ncap], 4 * ncap);
    for (auto& a : alphas) a = tr.draw();
    tr.absorb(&T[o_caps + 8 * ncap], 4 * ncap);
    const E zeta = tr.draw_ext();
    // FriOpenings (plonk/proof.rs:346-380): constants, sigmas, wires, zs, partial products, quotient at zeta; zs_next at g zeta
    tr.absorb(&T[o_const], 2 * d.num_constants); tr.absorb(&T[o_sig], 2 * R); tr.absorb(&T[o_wires], 2 * W); tr.absorb(&T[o_zs], 2 * nch);
    tr.absorb(&T[o_pp], 2 * nch * NPP); tr.absorb(&T[o_quot], 2 * nch * QF); tr.absorb(&T[o_lk], 2 * nch * NLP);
    tr.absorb(&T[o_zsn], 2 * nch); tr.absorb(&T[o_lkn], 2 * nch * NLP);
    const E fri_alpha = tr.draw_ext();
    E fri_betas[8];
    for (unsigned r = 0; r < d.num_fri_rounds; r++) { tr.absorb(&T[o_fcaps + (size_t)r * 4 * ncap], 4 * ncap); fri_betas[r] = tr.draw_ext(); }
    tr.absorb(&T[o_final], 2 * final_len);
    tr.absorb(pow_witness);
    const gl_t pow_response = tr.draw();
    std::vector<size_t> x_index(d.num_query_rounds);
    for (auto& x : x_index) x = (size_t)(tr.draw() % (uint64_t)N);

    // ---- vanishing(zeta) == Z_H(zeta) * t(zeta) per challenge (plonk/verifier.rs:64-101, vanishing_poly.rs:54-160) ----
    {
        std::vector<E> consts(d.num_constants), wires(W);
        for (size_t i = 0; i < d.num_constants; i++) consts[i] = ext_at(o_const, i);
        for (size_t i = 0; i < W; i++) wires[i] = ext_at(o_wires, i);
        const E zeta_n = e_pow2k(zeta, lgn), z_h = e_sub(zeta_n, e_of(1));
        // L_0(zeta) = (zeta^n - 1) / (n (zeta - 1))  (plonk_common.rs:61-71)
        const E l0 = e_eq(zeta, e_of(1)) ? e_of(1) : e_mul(z_h, gl2_inv(e_scale(e_sub(zeta, e_of(1)), (gl_t)n)));
        // gate constraints, summed slot-wise with each gate's selector filter (vanishing_poly.rs:671-699, gate.rs:277-284)
        const size_t NGC = 123;
        std::vector<E> gate_terms(NGC, e_of(0));
        const E* gate_consts = consts.data() + d.num_selectors + d.num_lookup_selectors;      // gate.rs:129-133
        E tmp[123];
        for (unsigned g = 0; g < d.num_gates; g++) {
            const E sel = consts[d.gate_selector_index[g]];
            E filter = e_of(1);
            for (unsigned i = d.gate_group_start[g]; i < d.gate_group_end[g]; i++) if (i != g) filter = e_mul(filter, e_sub(e_of(i), sel));
            if (d.num_selectors > 1) filter = e_mul(filter, e_sub(e_of(glhost::UNUSED_SELECTOR), sel));
            size_t cnt = 0;
            switch (d.gate_types[g]) {
                case glhost::G_NOOP: case glhost::G_LOOKUP: case glhost::G_LOOKUP_TABLE: break;          // no main-trace constraints (lookup.rs:72-75)
                case glhost::G_CONSTANT: cnt = 2; for (int i = 0; i < 2; i++) tmp[i] = e_sub(gate_consts[i], wires[i]); break;                        // constant.rs:59-66
                case glhost::G_PUBLIC_INPUT: cnt = 4; for (int i = 0; i < 4; i++) tmp[i] = e_sub(wires[i], e_of(pi_hash[i])); break;                     // public_input.rs:44-49
                case glhost::G_ARITHMETIC: cnt = 20;                                                                                                    // arithmetic_base.rs:72-92
                    for (int i = 0; i < 20; i++) tmp[i] = e_sub(wires[4 * i + 3], e_add(e_mul(e_mul(wires[4 * i], wires[4 * i + 1]), gate_consts[0]), e_mul(wires[4 * i + 2], gate_consts[1])));
                    break;
                case glhost::G_BASE_SUM: {                                                                                                              // base_sum.rs:63-76, B = 2
                    cnt = 1 + glhost::BASE_SUM_LIMBS;
                    E computed = e_of(0);                                                                                                                // reduce_with_powers(limbs, 2)
                    for (int i = glhost::BASE_SUM_LIMBS; i-- > 0;) computed = e_add(e_add(computed, computed), wires[1 + i]);
                    tmp[0] = e_sub(computed, wires[0]);
                    for (int i = 0; i < glhost::BASE_SUM_LIMBS; i++) tmp[1 + i] = e_mul(wires[1 + i], e_sub(wires[1 + i], e_of(1)));
                    break;
                }
                case glhost::G_RANDOM_ACCESS: {                                                                                                         // random_access.rs:139-184
                    const glhost::RandomAccessLayout ra(d.gate_params[g]);
                    for (uint32_t copy = 0; copy < ra.num_copies; copy++) {
                        std::vector<E> items(ra.vec_size);
                        for (uint32_t i = 0; i < ra.vec_size; i++) items[i] = wires[ra.wire_list_item(i, copy)];
                        for (uint32_t i = 0; i < ra.bits; i++) { const E b = wires[ra.wire_bit(i, copy)]; tmp[cnt++] = e_mul(b, e_sub(b, e_of(1))); }
                        E rec = e_of(0);
                        for (uint32_t i = ra.bits; i-- > 0;) rec = e_add(e_add(rec, rec), wires[ra.wire_bit(i, copy)]);
                        tmp[cnt++] = e_sub(rec, wires[ra.wire_access_index(copy)]);
                        for (uint32_t i = 0; i < ra.bits; i++) {                                                                                         // fold the list by bit i
                            const E b = wires[ra.wire_bit(i, copy)];
                            for (size_t j = 0; 2 * j + 1 < items.size(); j++) items[j] = e_add(items[2 * j], e_mul(b, e_sub(items[2 * j + 1], items[2 * j])));
                            items.resize(items.size() / 2);
                        }
                        tmp[cnt++] = e_sub(items[0], wires[ra.wire_claimed_element(copy)]);
                    }
                    for (uint32_t i = 0; i < ra.num_extra_constants; i++) tmp[cnt++] = e_sub(gate_consts[i], wires[ra.wire_extra_constant(i)]);
                    break;
                }
                case glhost::G_EXPONENTIATION: {                                                                                                        // exponentiation.rs:88-124
                    const int n = glhost::EXP_POWER_BITS;
                    cnt = n + 1;
                    for (int i = 0; i < n; i++) {                                                                                                        // square-and-multiply, bits big-endian
                        const E prev = i == 0 ? e_of(1) : e_mul(wires[2 + n + i - 1], wires[2 + n + i - 1]);
                        const E bit = wires[1 + (n - 1 - i)];
                        tmp[i] = e_sub(e_mul(prev, e_add(e_mul(bit, wires[0]), e_sub(e_of(1), bit))), wires[2 + n + i]);
                    }
                    tmp[n] = e_sub(wires[1 + n], wires[2 + n + n - 1]);
                    break;
                }
                default: cnt = 123; poseidon_gate_constraints(wires.data(), tmp); break;
            }
            for (size_t j = 0; j < cnt; j++) gate_terms[j] = e_add(gate_terms[j], e_mul(filter, tmp[j]));
        }
        for (size_t c = 0; c < nch; c++) {
            // terms in the order of vanishing_poly.rs:141-147: all L_0 (Z - 1), all partial-product checks, gate constraints
            std::vector<E> terms;
            for (size_t i = 0; i < nch; i++) terms.push_back(e_mul(l0, e_sub(ext_at(o_zs, i), e_of(1))));
            for (size_t i = 0; i < nch; i++) {
                E num[80], den[80];
                for (size_t j = 0; j < R; j++) {
                    const E wj = wires[j];
                    num[j] = e_add(e_add(wj, e_scale(e_scale(zeta, d.k_is[j]), betas[i])), e_of(gammas[i]));
                    den[j] = e_add(e_add(wj, e_scale(ext_at(o_sig, j), betas[i])), e_of(gammas[i]));
                }
                // check_partial_products (util/partial_products.rs:52-76): chunks of quotient_degree_factor wires
                const size_t chunks = R / QF;
                for (size_t k = 0; k < chunks; k++) {
                    const E prev = k == 0 ? ext_at(o_zs, i) : ext_at(o_pp, i * NPP + k - 1);
                    const E next = k == chunks - 1 ? ext_at(o_zsn, i) : ext_at(o_pp, i * NPP + k);
                    E np = e_of(1), dp = e_of(1);
                    for (size_t j = k * QF; j < (k + 1) * QF; j++) { np = e_mul(np, num[j]); dp = e_mul(dp, den[j]); }
                    terms.push_back(e_sub(e_mul(prev, np), e_mul(next, dp)));
                }
            }
            // lookup constraints of every challenge (vanishing_poly.rs:263-281, 337-500)
            for (size_t i = 0; i < nch && NLP; i++) {
                const gl_t* dl = deltas + glhost::NUM_COINS_LOOKUP * i;
                const size_t num_sldc = NLP - 1, lu_degree = QF - 1, lut_degree = (glhost::LOOKUP_TABLE_SLOTS + num_sldc - 1) / num_sldc;
                const E* sel = consts.data() + d.num_selectors;
                auto zx = [&](size_t k) { return ext_at(o_lk, i * NLP + 1 + k); };
                auto zgx = [&](size_t k) { return ext_at(o_lkn, i * NLP + 1 + k); };
                const E z_re = ext_at(o_lk, i * NLP), next_z_re = ext_at(o_lkn, i * NLP);
                E looked[glhost::LOOKUP_TABLE_SLOTS], combo[glhost::LOOKUP_TABLE_SLOTS], looking[glhost::LOOKUP_SLOTS];
                for (int sl = 0; sl < glhost::LOOKUP_TABLE_SLOTS; sl++) {
                    looked[sl] = e_add(wires[3 * sl], e_scale(wires[3 * sl + 1], dl[glhost::LU_CH_A]));
                    combo[sl] = e_add(wires[3 * sl], e_scale(wires[3 * sl + 1], dl[glhost::LU_CH_B]));
                }
                for (int sl = 0; sl < glhost::LOOKUP_SLOTS; sl++) looking[sl] = e_add(wires[2 * sl], e_scale(wires[2 * sl + 1], dl[glhost::LU_CH_A]));
                terms.push_back(e_mul(sel[glhost::LU_SEL_LAST_LDC], zx(num_sldc - 1)));
                terms.push_back(e_mul(sel[glhost::LU_SEL_INIT_SRE], zx(0)));
                terms.push_back(e_mul(sel[glhost::LU_SEL_INIT_SRE], z_re));
                // final RE, one per table: the table's polynomial at delta (get_lut_poly, vanishing_poly.rs:31-49), on the table's end selector
                for (unsigned t = 0; t < d.num_luts; t++)
                    terms.push_back(e_mul(sel[glhost::LU_SEL_START_END + t], e_sub(z_re, e_of(glhost::lut_poly_at_delta(d, t, dl[glhost::LU_CH_B], dl[glhost::LU_CH_DELTA])))));
                E cur = next_z_re;
                for (int sl = 0; sl < glhost::LOOKUP_TABLE_SLOTS; sl++) cur = e_add(e_scale(cur, dl[glhost::LU_CH_DELTA]), combo[sl]);
                terms.push_back(e_mul(sel[glhost::LU_SEL_TRANS_SRE], e_sub(z_re, cur)));
                const E al = e_of(dl[glhost::LU_CH_ALPHA]);
                for (size_t poly = 0; poly < num_sldc; poly++) {
                    const size_t t0 = poly * lut_degree, t1 = std::min<size_t>((poly + 1) * lut_degree, glhost::LOOKUP_TABLE_SLOTS);
                    const size_t u0 = poly * lu_degree, u1 = std::min<size_t>((poly + 1) * lu_degree, glhost::LOOKUP_SLOTS);
                    E lut_prod = e_of(1), lu_prod = e_of(1), lu_sum = e_of(0), lut_sum_mul = e_of(0);
                    for (size_t a = t0; a < t1; a++) lut_prod = e_mul(lut_prod, e_sub(al, looked[a]));
                    for (size_t a = u0; a < u1; a++) lu_prod = e_mul(lu_prod, e_sub(al, looking[a]));
                    for (size_t a = u0; a < u1; a++) { E pr = e_of(1); for (size_t b = u0; b < u1; b++) if (b != a) pr = e_mul(pr, e_sub(al, looking[b])); lu_sum = e_add(lu_sum, pr); }
                    for (size_t a = t0; a < t1; a++) { E pr = e_of(1); for (size_t b = t0; b < t1; b++) if (b != a) pr = e_mul(pr, e_sub(al, looked[b])); lut_sum_mul = e_add(lut_sum_mul, e_mul(wires[3 * a + 2], pr)); }
                    const E prev = poly == 0 ? zgx(num_sldc - 1) : zx(poly - 1);
                    terms.push_back(e_mul(sel[glhost::LU_SEL_TRANS_SRE], e_sub(e_mul(lut_prod, e_sub(zx(poly), prev)), lut_sum_mul)));
                    terms.push_back(e_mul(sel[glhost::LU_SEL_TRANS_LDC], e_add(e_mul(lu_prod, e_sub(zx(poly), prev)), lu_sum)));
                }
            }
            terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
            E acc = e_of(0);                                                      // reduce_with_powers (plonk_common.rs:97-128)
            for (size_t t = terms.size(); t-- > 0;) acc = e_add(terms[t], e_scale(acc, alphas[c]));
            E tz = e_of(0);
            for (size_t k = QF; k-- > 0;) tz = e_add(e_mul(tz, zeta_n), ext_at(o_quot, c * QF + k));
            if (!e_eq(acc, e_mul(z_h, tz))) return reject("vanishing polynomial identity fails at zeta");
        }
    }

    // ---- FRI (fri/verifier.rs:62-260) ----
    if (pow_response != 0 && (unsigned)__builtin_clzll(pow_response) < d.proof_of_work_bits) return reject("invalid proof of work witness");
    const E gzeta = e_scale(zeta, glhost::root_of_unity(lgn));
    auto horner = [&](const std::vector<E>& v) { E acc = e_of(0); for (size_t i = v.size(); i-- > 0;) acc = e_add(e_mul(acc, fri_alpha), v[i]); return acc; };
    // PrecomputedReducedOpenings (fri/verifier.rs:243-260)
    std::vector<E> open0, open1;
    for (size_t i = 0; i < d.num_constants; i++) open0.push_back(ext_at(o_const, i));
    for (size_t i = 0; i < R; i++) open0.push_back(ext_at(o_sig, i));
    for (size_t i = 0; i < W; i++) open0.push_back(ext_at(o_wires, i));
    for (size_t i = 0; i < nch; i++) open0.push_back(ext_at(o_zs, i));
    for (size_t i = 0; i < nch * NPP; i++) open0.push_back(ext_at(o_pp, i));
    for (size_t i = 0; i < nch * QF; i++) open0.push_back(ext_at(o_quot, i));
    for (size_t i = 0; i < nch * NLP; i++) open0.push_back(ext_at(o_lk, i));          // lookup polynomials come last in both batches (circuit_data.rs:564-597)
    for (size_t i = 0; i < nch; i++) open1.push_back(ext_at(o_zsn, i));
    for (size_t i = 0; i < nch * NLP; i++) open1.push_back(ext_at(o_lkn, i));
    const E red0 = horner(open0), red1 = horner(open1);
    E alpha_shift = e_of(1);
    for (size_t i = 0; i < open1.size(); i++) alpha_shift = e_mul(alpha_shift, fri_alpha);
    const gl_t* caps[4] = {constants_sigmas_cap, &T[o_caps], &T[o_caps + 4 * ncap], &T[o_caps + 8 * ncap]};
    const gl_t wN = glhost::root_of_unity(lgN);
    std::vector<E> ev0, ev1;
    for (unsigned q = 0; q < d.num_query_rounds; q++) {
        size_t x = x_index[q];
        for (int o = 0; o < 4; o++) {
            const PathRef& pr = init_paths[q * 4 + o];
            if (!path_opens_to_cap(&T[pr.leaf], pr.leaf_len, x, &T[pr.sib], pr.nsib, caps[o], ncap)) return reject("initial Merkle proof fails");
        }
        // subgroup_x = g * w_N^{reverse_bits(x_index)} (fri/verifier.rs:183-186)
        size_t rev = 0;
        for (unsigned i = 0; i < lgN; i++) rev |= ((x >> i) & 1) << (lgN - 1 - i);
        gl_t subgroup_x = gl_canon(gl_mul(GL_MULT_GENERATOR, gl_exp(wN, rev)));
        // fri_combine_initial (fri/verifier.rs:124-165): batch 0 = every polynomial at zeta, batch 1 = the Z polynomials at g zeta
        ev0.clear(); ev1.clear();
        for (int o = 0; o < 4; o++) { const PathRef& pr = init_paths[q * 4 + o]; for (size_t i = 0; i < (o == 2 ? nzp : pr.leaf_len); i++) ev0.push_back(e_of(T[pr.leaf + i])); }
        for (size_t i = nzp; i < widths[2]; i++) ev0.push_back(e_of(T[init_paths[q * 4 + 2].leaf + i]));
        for (size_t i = 0; i < nch; i++) ev1.push_back(e_of(T[init_paths[q * 4 + 2].leaf + i]));
        for (size_t i = nzp; i < widths[2]; i++) ev1.push_back(e_of(T[init_paths[q * 4 + 2].leaf + i]));
        const E sx = e_of(subgroup_x);
        E eval = e_mul(e_sub(horner(ev0), red0), gl2_inv(e_sub(sx, zeta)));
        eval = e_add(e_mul(eval, alpha_shift), e_mul(e_sub(horner(ev1), red1), gl2_inv(e_sub(sx, gzeta))));
        for (unsigned r = 0; r < d.num_fri_rounds; r++) {
            const unsigned ab = d.fri_arity_bits[r];
            const size_t arity = size_t(1) << ab, coset = x >> ab, within = x & (arity - 1);
            const PathRef& pr = step_paths[q * d.num_fri_rounds + r];
            if (!e_eq(ext_at(pr.leaf, within), eval)) return reject("FRI consistency check fails");
            // compute_evaluation (fri/verifier.rs:21-47): the degree < arity interpolant through the coset, at beta
            const gl_t g = glhost::root_of_unity(ab);
            size_t wrev = 0;
            for (unsigned i = 0; i < ab; i++) wrev |= ((within >> i) & 1) << (ab - 1 - i);
            const gl_t start = gl_canon(gl_mul(subgroup_x, gl_exp(g, arity - wrev)));      // coset_start = x * g^{-rev(within)}
            std::vector<gl_t> pts(arity);
            { gl_t y = 1; for (size_t k = 0; k < arity; k++) { pts[k] = gl_canon(gl_mul(start, y)); y = gl_mul(y, g); } }
            E acc = e_of(0);
            for (size_t a = 0; a < arity; a++) {
                size_t arev = 0;
                for (unsigned i = 0; i < ab; i++) arev |= ((a >> i) & 1) << (ab - 1 - i);      // evals are stored in bit-reversed order
                E numer = e_of(1); gl_t denom = 1;
                for (size_t b = 0; b < arity; b++) if (b != a) { numer = e_mul(numer, e_sub(fri_betas[r], e_of(pts[b]))); denom = gl_mul(denom, gl_sub(pts[a], pts[b])); }
                acc = e_add(acc, e_mul(ext_at(pr.leaf, arev), e_scale(numer, gl_inv(denom))));
            }
            eval = acc;
            if (!path_opens_to_cap(&T[pr.leaf], pr.leaf_len, coset, &T[pr.sib], pr.nsib, &T[o_fcaps + (size_t)r * 4 * ncap], ncap)) return reject("FRI step Merkle proof fails");
            for (unsigned i = 0; i < ab; i++) subgroup_x = gl_sqr(subgroup_x);
            x = coset;
        }
        E fin = e_of(0);
        const E sxf = e_of(gl_canon(subgroup_x));
        for (size_t i = final_len; i-- > 0;) fin = e_add(e_mul(fin, sxf), ext_at(o_final, i));
        if (!e_eq(fin, eval)) return reject("final polynomial evaluation is invalid");
    }
    return GL_OK;
}

// convenience for callers that hold the host circuit: CircuitData::verify (plonk/circuit_data.rs:153-155)
extern "C" int gl_host_circuit_verify(const gl_host_circuit* hc, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4],
                                      const uint8_t* proof_bytes, size_t num_bytes) {
    GL_REQUIRE(hc, GL_ERR_ARG, "gl_host_circuit_verify: null circuit");
    return gl_verify(&hc->hc.desc, constants_sigmas_cap, circuit_digest, proof_bytes, num_bytes);
}
