// Host-side (CPU, C++) circuit description for the matrix-multiplication demo family: the part of the reference's
// Rust host code that feeds the hot path -- CircuitBuilder + build() for this circuit
// (plonky2/src/bin/matrix_mul.rs:25-67, plonk/circuit_builder.rs:913-1146) and witness generation
// (plonk/prover.rs:118-133).  The Rust toolchain is absent, so the host layer above the C ABI is C++.
//
// This is NOT a generic CircuitBuilder: the matmul circuit's gate placement is derived in closed form from the
// order in which the demo issues its mul/add operations (each ArithmeticGate row holds 20 operations of one
// (const_0, const_1) kind, rows are opened on demand: gadgets/arithmetic.rs:87-99, circuit_builder.rs:665-695),
// and the copy-constraint classes are enumerated directly instead of through a union-find forest.
#pragma once
#include <stdint.h>
#include <mutex>
#include <string.h>
#include <algorithm>
#include <array>
#include <string>
#include <vector>
#include "gl64.cuh"
#include "poseidon.cuh"
#include "../../include/plonky2_mi355x.h"

namespace glhost {

static const gl_t MULT_GEN = 7;
static const gl_t POW2_GEN = 1753635133440165772ULL;
inline gl_t root_of_unity(unsigned lg) { gl_t r = POW2_GEN; for (unsigned i = lg; i < 32; i++) r = gl_sqr(r); return gl_canon(r); }

// gate type codes shared with the device kernels (order = the reference's sort by (degree, id))
enum { G_NOOP = 0, G_CONSTANT = 1, G_PUBLIC_INPUT = 2, G_ARITHMETIC = 3, G_POSEIDON = 4, G_BASE_SUM = 5, G_LOOKUP = 6, G_LOOKUP_TABLE = 7, G_EXPONENTIATION = 8,
       G_RANDOM_ACCESS = 9, G_LAST = G_RANDOM_ACCESS };
// RandomAccessGate::new_from_config(standard_recursion_config, bits) (gates/random_access.rs:55-110), bits = gate_params[g] in 1..6: copy c
// owns wires (2 + 2^bits) c ..: access index, claimed element, the list; then the extra constants; then (unrouted) every copy's index bits
struct RandomAccessLayout {
    uint32_t bits, vec_size, num_copies, num_extra_constants;
    GL_HD explicit RandomAccessLayout(uint32_t b) : bits(b), vec_size(1u << b) {
        const uint32_t by_routed = 80u / (2u + vec_size), by_wires = 135u / (2u + vec_size + bits);
        num_copies = by_routed < by_wires ? by_routed : by_wires;
        const uint32_t left = 80u - (2u + vec_size) * num_copies;
        num_extra_constants = left < 2u ? left : 2u;
    }
    GL_HD uint32_t wire_access_index(uint32_t c) const { return (2u + vec_size) * c; }
    GL_HD uint32_t wire_claimed_element(uint32_t c) const { return (2u + vec_size) * c + 1u; }
    GL_HD uint32_t wire_list_item(uint32_t i, uint32_t c) const { return (2u + vec_size) * c + 2u + i; }
    GL_HD uint32_t wire_extra_constant(uint32_t i) const { return (2u + vec_size) * num_copies + i; }
    GL_HD uint32_t wire_bit(uint32_t i, uint32_t c) const { return (2u + vec_size) * num_copies + num_extra_constants + c * bits + i; }
    GL_HD uint32_t num_constraints() const { return num_copies * (bits + 2u) + num_extra_constants; }      // random_access.rs:285-288
};
// ExponentiationGate::new_from_config (gates/exponentiation.rs:43-53): min(routed - 2, (wires - 2) / 2) = 66 power bits; wires: 0 base,
// 1..66 power bits (little-endian), 67 output, 68..133 intermediate values; 67 constraints of degree 4
enum { EXP_POWER_BITS = 66 };
enum { LOOKUP_SLOTS = 40, LOOKUP_TABLE_SLOTS = 26, NUM_COINS_LOOKUP = 4 };      // gates/lookup.rs:41-44, gates/lookup_table.rs:47-50, circuit_builder.rs:56-58
enum { LU_CH_A = 0, LU_CH_B = 1, LU_CH_ALPHA = 2, LU_CH_DELTA = 3 };            // LookupChallenges (circuit_builder.rs:61-71)
enum { LU_SEL_TRANS_SRE = 0, LU_SEL_TRANS_LDC = 1, LU_SEL_INIT_SRE = 2, LU_SEL_LAST_LDC = 3, LU_SEL_START_END = 4 };      // gates/selectors.rs:34-40
enum { BASE_SUM_LIMBS = 63 };      // BaseSumGate::<2>::new_from_config (gates/base_sum.rs:31-35): wire 0 = sum, wires 1..=63 = limbs
static const uint64_t UNUSED_SELECTOR = 0xFFFFFFFFull;        // gates/selectors.rs:14

// ---- lookup tables of a description: `lut` holds the tables one after the other ----
inline uint32_t lut_offset(const gl_circuit_desc& d, unsigned t) { uint32_t o = 0; for (unsigned i = 0; i < t && i < GL_MAX_LUTS; i++) o += d.lut_len[i]; return o; }
inline uint32_t lut_rows(const gl_circuit_desc& d, unsigned t) { return (d.lut_len[t] + LOOKUP_TABLE_SLOTS - 1) / LOOKUP_TABLE_SLOTS; }
// gate_params of the gates that are not lookup gates: 1..6 bits for a RandomAccessGate, 0 otherwise; nullptr = fine
inline const char* gate_params_error(const gl_circuit_desc& d) {
    for (unsigned g = 0; g < d.num_gates && g < GL_MAX_GATES; g++) {
        if (d.gate_types[g] == G_RANDOM_ACCESS) { if (d.gate_params[g] < 1 || d.gate_params[g] > 6) return "RandomAccessGate: gate_params must hold 1..6 bits"; }
        else if (d.gate_types[g] != G_LOOKUP && d.gate_types[g] != G_LOOKUP_TABLE && d.gate_params[g] != 0) return "bad gate_params";
    }
    return nullptr;
}
// the counts every consumer of a description relies on before it indexes lut / lut_len / the per-table rows; nullptr = fine
inline const char* lookup_shape_error(const gl_circuit_desc& d) {
    if (d.num_luts == 0) {
        if (d.num_lookup_polys || d.num_lookup_selectors) return "lookup polynomials / selectors without a lookup table";
        for (unsigned g = 0; g < d.num_gates && g < GL_MAX_GATES; g++)
            if (d.gate_types[g] == G_LOOKUP || d.gate_types[g] == G_LOOKUP_TABLE) return "lookup gates without a lookup table";
        return gate_params_error(d);
    }
    if (d.num_luts > GL_MAX_LUTS) return "more than GL_MAX_LUTS lookup tables";
    if (d.num_lookup_polys != 7 || d.num_lookup_selectors != LU_SEL_START_END + d.num_luts)
        return "lookups: 7 lookup polynomials and 4 + num_luts lookup selectors per challenge";
    uint32_t total = 0;
    for (unsigned t = 0; t < d.num_luts; t++) {
        if (d.lut_len[t] < 1 || d.lut_len[t] > GL_MAX_LUT_ENTRIES) return "a lookup table needs 1 .. GL_MAX_LUT_ENTRIES entries";
        total += d.lut_len[t];
    }
    if (total > GL_MAX_LUT_ENTRIES) return "the lookup tables together exceed GL_MAX_LUT_ENTRIES entries";
    for (unsigned g = 0; g < d.num_gates && g < GL_MAX_GATES; g++)
        if ((d.gate_types[g] == G_LOOKUP || d.gate_types[g] == G_LOOKUP_TABLE) ? d.gate_params[g] >= d.num_luts : (d.gate_types[g] != G_RANDOM_ACCESS && d.gate_params[g] != 0)) return "bad gate_params";
    return gate_params_error(d);
}
// get_lut_poly(common_data, t, deltas, 26 * rows).eval(delta) (vanishing_poly.rs:31-49): the table's combos inp + b out, zero-padded to
// whole LookupTableGate rows and REVERSED, as coefficients of a polynomial at the delta challenge (Horner from combo_0)
inline gl_t lut_poly_at_delta(const gl_circuit_desc& d, unsigned t, gl_t b, gl_t delta) {
    const uint16_t* lut = d.lut + 2 * (size_t)lut_offset(d, t);
    const size_t deg = (size_t)lut_rows(d, t) * LOOKUP_TABLE_SLOTS;
    gl_t f = 0;
    for (size_t k = 0; k < deg; k++) f = gl_add(gl_mul(f, delta), k < d.lut_len[t] ? gl_add((gl_t)lut[2 * k], gl_mul(b, (gl_t)lut[2 * k + 1])) : (gl_t)0);
    return gl_canon(f);
}

// PoseidonGate wire layout (gates/poseidon.rs:36-96)
enum { PW_INPUT = 0, PW_OUTPUT = 12, PW_SWAP = 24, PW_DELTA = 25, PW_FULL0 = 29, PW_PARTIAL = 65, PW_FULL1 = 87, PW_END = 135 };

inline void host_hash_no_pad(const gl_t* in, size_t n, gl_t* out4) {     // hash/hashing.rs:117-146
    gl_t s[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t c = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < c; i++) s[i] = in[off + i];
        psd_permute(s);
    }
    for (int i = 0; i < 4; i++) out4[i] = gl_canon(s[i]);
}

struct HostCircuit {
    gl_circuit_desc desc;
    size_t m = 0, n = 0;
    std::vector<uint8_t> row_gate;                     // gate type per row
    std::vector<gl_t> constants_sigmas;                // column-major [num_constants + 80][n] VALUES; the sigma part is filled on demand
    std::vector<uint64_t> wire_class;                  // [80][n]: copy-constraint class of every routed wire (equal id = constrained equal)
    mutable std::once_flag sigmas_once;                // host sigma values are only needed by callers that ask for the columns
    void ensure_host_sigmas() const;
    // witness recipe
    size_t first_poseidon_row = 0, num_poseidon_rows = 0, pi_row = 0, constant_row = 0;
    std::vector<uint32_t> mul_row, add_row;            // row of the t-th mul / add row-block (20 ops each)
};

// position (row, slot) of the t-th operation of a kind
struct OpPos { uint32_t row, slot; };

inline int build_matmul(size_t m, HostCircuit* hc) {
    if (m < 1 || m > 256) return GL_ERR_ARG;
    hc->m = m;
    const size_t n_mul = m * m * m, n_add = m * m * (m - 1);
    // --- row allocation: replay the demo's op order; a row is opened when a kind has no free slot ---
    // per (i,j): k = 0: mul; k >= 1: mul, add
    std::vector<uint32_t>& mul_row = hc->mul_row; std::vector<uint32_t>& add_row = hc->add_row;
    mul_row.clear(); add_row.clear();
    {
        size_t mc = 0, ac = 0; uint32_t rows = 0;
        for (size_t ij = 0; ij < m * m; ij++)
            for (size_t k = 0; k < m; k++) {
                if (mc % 20 == 0) mul_row.push_back(rows++);
                mc++;
                if (k >= 1) { if (ac % 20 == 0) add_row.push_back(rows++); ac++; }
            }
        (void)n_mul; (void)n_add;
        hc->first_poseidon_row = rows;
    }
    const size_t n_pi = 3 * m * m;
    hc->num_poseidon_rows = (n_pi + 7) / 8;
    hc->pi_row = hc->first_poseidon_row + hc->num_poseidon_rows;
    hc->constant_row = hc->pi_row + 1;
    const size_t rows_used = hc->constant_row + 1;
    unsigned lg = 0; while ((size_t(1) << lg) < rows_used) lg++;
    const size_t n = size_t(1) << lg;
    hc->n = n;
    hc->row_gate.assign(n, G_NOOP);
    for (auto r : mul_row) hc->row_gate[r] = G_ARITHMETIC;
    for (auto r : add_row) hc->row_gate[r] = G_ARITHMETIC;
    for (size_t r = 0; r < hc->num_poseidon_rows; r++) hc->row_gate[hc->first_poseidon_row + r] = G_POSEIDON;
    hc->row_gate[hc->pi_row] = G_PUBLIC_INPUT;
    hc->row_gate[hc->constant_row] = G_CONSTANT;

    // --- descriptor (standard_recursion_config, circuit_data.rs:72-90) ---
    gl_circuit_desc& d = hc->desc;
    ::memset((void*)&d, 0, sizeof d);
    d.degree_bits = lg; d.num_wires = 135; d.num_routed_wires = 80; d.num_challenges = 2; d.quotient_degree_factor = 8;
    d.rate_bits = 3; d.cap_height = 4; d.proof_of_work_bits = 16; d.num_query_rounds = 28;
    d.num_public_inputs = (uint32_t)n_pi;
    {   // ConstantArityBits(4, 5) (fri/reduction_strategies.rs:39-49)
        unsigned db = lg; d.num_fri_rounds = 0;
        while (db > 5 && db + d.rate_bits - 4 >= d.cap_height) { d.fri_arity_bits[d.num_fri_rounds++] = 4; db -= 4; }
    }
    // gates present, sorted by (degree, id): Noop(0) < Constant(1) < PublicInput(1) < Arithmetic(3) < Poseidon(7)
    bool present[5] = {false, false, false, false, false};
    for (auto g : hc->row_gate) present[g] = true;
    const unsigned degree_of[5] = {0, 1, 1, 3, 7};
    d.num_gates = 0;
    for (int g = 0; g < 5; g++) if (present[g]) d.gate_types[d.num_gates++] = (uint8_t)g;
    // selector groups (gates/selectors.rs:110-185), max_degree = quotient_degree_factor + 1
    const unsigned max_degree = d.quotient_degree_factor + 1, ng = d.num_gates;
    d.num_selectors = 0;
    if (degree_of[d.gate_types[ng - 1]] + ng - 1 <= max_degree) {
        d.num_selectors = 1;
        for (unsigned i = 0; i < ng; i++) { d.gate_selector_index[i] = 0; d.gate_group_start[i] = 0; d.gate_group_end[i] = ng; }
    } else {
        unsigned start = 0;
        while (start < ng) {
            unsigned size = 0;
            while (start + size < ng && size + degree_of[d.gate_types[start + size]] < max_degree) size++;
            for (unsigned i = start; i < start + size; i++) { d.gate_selector_index[i] = d.num_selectors; d.gate_group_start[i] = start; d.gate_group_end[i] = start + size; }
            d.num_selectors++;
            start += size;
        }
    }
    d.num_constants = d.num_selectors + 2;             // selectors + the two gate-constant columns
    { gl_t x = 1; for (int j = 0; j < 80; j++) { d.k_is[j] = x; x = gl_canon(gl_mul(x, MULT_GEN)); } }   // field/src/cosets.rs:9-24

    // --- constants: selector columns + gate constants ---
    const size_t ncs = d.num_constants + 80;
    hc->constants_sigmas.assign(ncs * n, 0);
    gl_t* cs = hc->constants_sigmas.data();
    unsigned gate_index_of[5] = {0, 0, 0, 0, 0};
    for (unsigned i = 0; i < ng; i++) gate_index_of[d.gate_types[i]] = i;
    for (size_t r = 0; r < n; r++) {
        unsigned gi = gate_index_of[hc->row_gate[r]];
        for (unsigned s = 0; s < d.num_selectors; s++)
            cs[s * n + r] = (d.num_selectors == 1 || d.gate_selector_index[gi] == s) ? gi : UNUSED_SELECTOR;
    }
    gl_t* c0 = cs + (size_t)d.num_selectors * n; gl_t* c1 = c0 + n;
    for (auto r : mul_row) { c0[r] = 1; c1[r] = 0; }            // x*y = 1*x*y + 0*x  (arithmetic.rs:210-213)
    for (auto r : add_row) { c0[r] = 1; c1[r] = 1; }            // x+y = 1*x*1 + 1*y  (arithmetic.rs:187-191)
    c0[hc->constant_row] = 0; c1[hc->constant_row] = 1;         // constants {0, 1} sorted by value (circuit_builder.rs:946-959)

    // --- copy-constraint classes -> sigma (permutation_argument.rs:85-170) ---
    // Every routed wire gets a class key; wires with equal keys are one partition subset.  Keys:
    enum : uint64_t { K_A = 1ull << 60, K_B = 2ull << 60, K_PROD = 3ull << 60, K_SUM = 4ull << 60, K_ZERO = 5ull << 60,
                      K_ONE = 6ull << 60, K_HOUT = 7ull << 60, K_SELF = 8ull << 60 };
    std::vector<uint64_t>& key = hc->wire_class;
    key.assign(80 * n, 0);
    for (size_t r = 0; r < n; r++) for (size_t c = 0; c < 80; c++) key[c * n + r] = K_SELF | (c * n + r);
    auto setk = [&](size_t row, size_t col, uint64_t k) { if (col < 80) key[col * n + row] = k; };
    // public-input target t (order a_ij, b_ij, c_ij per (i,j)): key of the value it carries
    auto c_key = [&](size_t i, size_t j) -> uint64_t { return m == 1 ? (K_PROD | ((i * m + j) * m)) : (K_SUM | ((i * m + j) * m + (m - 1))); };
    {
        size_t mc = 0, ac = 0;
        for (size_t i = 0; i < m; i++)
            for (size_t j = 0; j < m; j++)
                for (size_t k = 0; k < m; k++) {
                    size_t row = mul_row[mc / 20], s = mc % 20; mc++;
                    uint64_t ka = K_A | (i * m + k), kb = K_B | (k * m + j), kp = K_PROD | ((i * m + j) * m + k);
                    setk(row, 4 * s, ka); setk(row, 4 * s + 1, kb); setk(row, 4 * s + 2, ka); setk(row, 4 * s + 3, kp);
                    if (k >= 1) {
                        size_t arow = add_row[ac / 20], as = ac % 20; ac++;
                        uint64_t prev = (k == 1) ? (K_PROD | ((i * m + j) * m)) : (K_SUM | ((i * m + j) * m + (k - 1)));
                        setk(arow, 4 * as, prev); setk(arow, 4 * as + 1, K_ONE); setk(arow, 4 * as + 2, kp);
                        setk(arow, 4 * as + 3, K_SUM | ((i * m + j) * m + k));
                    }
                }
    }
    // Poseidon rows: state starts as [zero; 12]; each chunk overwrites the first c entries with PI targets
    {
        std::array<uint64_t, 12> state; state.fill(K_ZERO);
        for (size_t pr = 0; pr < hc->num_poseidon_rows; pr++) {
            size_t row = hc->first_poseidon_row + pr, off = pr * 8, c = std::min<size_t>(8, n_pi - off);
            for (size_t t = 0; t < c; t++) {
                size_t pi = off + t, ij = pi / 3, which = pi % 3, i = ij / m, j = ij % m;
                state[t] = which == 0 ? (K_A | (i * m + j)) : which == 1 ? (K_B | (i * m + j)) : c_key(i, j);
            }
            setk(row, PW_SWAP, K_ZERO);
            for (int t = 0; t < 12; t++) setk(row, PW_INPUT + t, state[t]);
            for (int t = 0; t < 12; t++) { state[t] = K_HOUT | (pr * 12 + t); setk(row, PW_OUTPUT + t, state[t]); }
        }
        for (int t = 0; t < 4; t++) setk(hc->pi_row, t, state[t]);
    }
    setk(hc->constant_row, 0, K_ZERO); setk(hc->constant_row, 1, K_ONE);
    return GL_OK;
}

inline uint64_t splitmix64_next(uint64_t& x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// PoseidonGenerator (gates/poseidon.rs:430-497): all wires of one PoseidonGate row with swap = 0
inline void poseidon_row_witness(const gl_t* in12, gl_t* row /* 135, stride `stride` */, size_t stride) {
    gl_t s[12];
    for (int i = 0; i < 12; i++) { s[i] = in12[i]; row[(PW_INPUT + i) * stride] = gl_canon(in12[i]); }
    row[PW_SWAP * stride] = 0;
    for (int i = 0; i < 4; i++) row[(PW_DELTA + i) * stride] = 0;          // swap * (rhs - lhs)
    int round = 0;
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], POSEIDON_RC[12 * round + i]);
        if (r != 0) for (int i = 0; i < 12; i++) row[(PW_FULL0 + 12 * (r - 1) + i) * stride] = gl_canon(s[i]);
        for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
        psd_mds(s);
        round++;
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], POSEIDON_PARTIAL_FIRST_RC[i]);
    {
        gl_t t[12]; t[0] = s[0];
        for (int c = 1; c < 12; c++) { PsdHostDot d; for (int r = 1; r < 12; r++) psd_host_dot_term(d, s[r], POSEIDON_PARTIAL_INIT[(r - 1) * 11 + (c - 1)]); t[c] = psd_host_dot_reduce(d); }
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        row[(PW_PARTIAL + r) * stride] = gl_canon(s[0]);
        gl_t s0 = gl_add_c(psd_sbox(s[0]), POSEIDON_PARTIAL_RC[r]);
        PsdHostDot dd;
        psd_host_dot_term(dd, s0, 25);
        for (int i = 1; i < 12; i++) psd_host_dot_term(dd, s[i], POSEIDON_PARTIAL_ROW[r * 11 + i - 1]);
        for (int i = 1; i < 12; i++) s[i] = gl_mul_add(s[i], s0, POSEIDON_PARTIAL_COL[r * 11 + i - 1]);
        s[0] = psd_host_dot_reduce(dd);
    }
    round += POSEIDON_PARTIAL_ROUNDS;
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add_c(s[i], POSEIDON_RC[12 * round + i]);
        for (int i = 0; i < 12; i++) row[(PW_FULL1 + 12 * r + i) * stride] = gl_canon(s[i]);
        for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
        psd_mds(s);
        round++;
    }
    for (int i = 0; i < 12; i++) row[(PW_OUTPUT + i) * stride] = gl_canon(s[i]);
}

// Full witness matrix wires[135][n] (column-major, canonical) and the public inputs, computed directly.
inline int matmul_witness(const HostCircuit& hc, const gl_t* a, const gl_t* b, uint64_t filler_seed, gl_t* wires, gl_t* pis) {
    const size_t m = hc.m, n = hc.n;
    std::fill(wires, wires + 135 * n, gl_t(0));
    auto W = [&](size_t row, size_t col) -> gl_t& { return wires[col * n + row]; };
    std::vector<gl_t> cvals(m * m);
    size_t mc = 0, ac = 0;
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < m; j++) {
            gl_t cur = 0;
            for (size_t k = 0; k < m; k++) {
                size_t row = hc.mul_row[mc / 20], s = mc % 20; mc++;
                gl_t x = gl_canon(a[i * m + k]), y = gl_canon(b[k * m + j]), p = gl_canon(gl_mul(x, y));
                W(row, 4 * s) = x; W(row, 4 * s + 1) = y; W(row, 4 * s + 2) = x; W(row, 4 * s + 3) = p;
                if (k == 0) { cur = p; continue; }
                size_t arow = hc.add_row[ac / 20], as = ac % 20; ac++;
                gl_t sum = gl_canon(gl_add(cur, p));
                W(arow, 4 * as) = cur; W(arow, 4 * as + 1) = 1; W(arow, 4 * as + 2) = p; W(arow, 4 * as + 3) = sum;
                cur = sum;
            }
            cvals[i * m + j] = cur;
        }
    const size_t n_pi = 3 * m * m;
    for (size_t ij = 0; ij < m * m; ij++) { pis[3 * ij] = gl_canon(a[ij]); pis[3 * ij + 1] = gl_canon(b[ij]); pis[3 * ij + 2] = cvals[ij]; }
    gl_t state[12] = {0};
    for (size_t pr = 0; pr < hc.num_poseidon_rows; pr++) {
        size_t off = pr * 8, c = std::min<size_t>(8, n_pi - off);
        for (size_t t = 0; t < c; t++) state[t] = pis[off + t];
        size_t row = hc.first_poseidon_row + pr;
        poseidon_row_witness(state, wires + row, n);
        for (int t = 0; t < 12; t++) state[t] = W(row, PW_OUTPUT + t);
    }
    for (int t = 0; t < 4; t++) W(hc.pi_row, t) = state[t];
    { uint64_t st = filler_seed; for (size_t c = 4; c < 135; c++) W(hc.pi_row, c) = splitmix64_next(st) % GL_P; }   // circuit_builder.rs:904-910
    W(hc.constant_row, 0) = 0; W(hc.constant_row, 1) = 1;
    return GL_OK;
}


// sigma VALUES on the host from the wire classes (permutation_argument.rs:85-170, circuit_builder.rs:1007-1014): every class is a
// cycle through its wires in (row, column) order; sigma(wire) = k_col(next) * w^row(next).  The device does the same in
// sigma.hip; this host form serves gl_host_circuit_constants_sigmas and the tests.
inline void HostCircuit::ensure_host_sigmas() const {
    std::call_once(sigmas_once, [this] {
        HostCircuit* hc = const_cast<HostCircuit*>(this);
        const gl_circuit_desc& d = hc->desc;
        const size_t n = hc->n;
        const unsigned lg = d.degree_bits;
        const std::vector<uint64_t>& key = hc->wire_class;
        gl_t* cs = hc->constants_sigmas.data();
        // sort wire positions by (key, row, col); the reference orders a subset by (row, column)
        std::vector<uint32_t> order(80 * n);
        for (size_t p = 0; p < order.size(); p++) order[p] = (uint32_t)p;
        auto rowcol = [&](uint32_t p) { return (uint64_t)(p % n) * 80 + p / n; };
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] != key[y] ? key[x] < key[y] : rowcol(x) < rowcol(y); });
        std::vector<gl_t> subgroup(n);
        { gl_t g = root_of_unity(lg), x = 1; for (size_t i = 0; i < n; i++) { subgroup[i] = x; x = gl_canon(gl_mul(x, g)); } }
        gl_t* sig = cs + (size_t)d.num_constants * n;
        for (size_t s = 0; s < order.size();) {
            size_t e = s;
            while (e < order.size() && key[order[e]] == key[order[s]]) e++;
            for (size_t q = s; q < e; q++) {
                uint32_t me = order[q], nb = order[q + 1 < e ? q + 1 : s];
                sig[me] = gl_canon(gl_mul(d.k_is[nb / n], subgroup[nb % n]));      // sigma(me) = k_col(nb) * w^row(nb)
            }
            s = e;
        }
    });
}
}  // namespace glhost

struct gl_host_circuit { glhost::HostCircuit hc; };     // the opaque handle of include/plonky2_mi355x.h
