// ORACLE (test infrastructure only).  CPU restatement of the gate constraint formulas the matmul demo circuit
// instantiates, generic over the evaluation field K (K = u64 for the prover's base-field LDE points,
// K = Ext2 for the verifier's point zeta).
// Follows:
//   plonky2/src/gates/arithmetic_base.rs:72-92,163-181   (ArithmeticGate, 20 ops per row)
//   plonky2/src/gates/poseidon.rs:36-96,113-272          (PoseidonGate wire layout + 123 constraints)
//   plonky2/src/gates/public_input.rs:97-110, constant.rs:59-66, noop.rs
//   plonky2/src/gates/gate.rs:121-146,277-284            (filter = prod_{i in group, i != row}(i - s) * (UNUSED - s))
//   plonky2/src/gates/selectors.rs:14,110-185            (selector groups)
//   plonky2/src/hash/poseidon.rs:200-214,264-274,311-366,429-450,495-503,552-559 (field-generic layers)
#pragma once
#include "gl_poseidon.hpp"
#include <string>

namespace orc {

// ---- field-generic helpers -------------------------------------------------------------------------
static inline u64 kadd(u64 a, u64 b) { return add(a, b); }
static inline u64 ksub(u64 a, u64 b) { return sub(a, b); }
static inline u64 kmul(u64 a, u64 b) { return mul(a, b); }
static inline u64 kscal(u64 a, u64 s) { return mul(a, s); }
static inline Ext2 kadd(Ext2 a, Ext2 b) { return eadd(a, b); }
static inline Ext2 ksub(Ext2 a, Ext2 b) { return esub(a, b); }
static inline Ext2 kmul(Ext2 a, Ext2 b) { return emul(a, b); }
static inline Ext2 kscal(Ext2 a, u64 s) { return escalar(a, s); }
template <class K> static inline K kconst(u64 c);
template <> inline u64 kconst<u64>(u64 c) { return c; }
template <> inline Ext2 kconst<Ext2>(u64 c) { return Ext2{c, 0}; }
template <class K> static inline K ksbox(K x) { K x2 = kmul(x, x), x4 = kmul(x2, x2), x3 = kmul(x, x2); return kmul(x3, x4); }

// (the numbering is the product's gl_circuit_desc.gate_types: 5 = BaseSumGate<2> with the 63 limbs of new_from_config)
//  6 = LookupGate (40 (input, output) slots), 7 = LookupTableGate (26 (input, output, multiplicity) slots): one of each per lookup table
//  8 = ExponentiationGate with the 66 power bits of new_from_config (gates/exponentiation.rs:43-53: min(routed - 2, (wires - 2) / 2))
//  9 = RandomAccessGate::new_from_config(config, bits) (gates/random_access.rs:55-72), one gate type per `bits` (1..6)
enum GateType { GATE_NOOP = 0, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON, GATE_BASE_SUM, GATE_LOOKUP, GATE_LOOKUP_TABLE, GATE_EXPONENTIATION,
                GATE_RANDOM_ACCESS, GATE_NUM_TYPES };
// RandomAccessGate layout for standard_recursion_config (135 wires, 80 routed, 2 constants): copy c owns wires (2 + 2^bits) c ..: access
// index, claimed element, the list; then the extra constants; then (unrouted) the index bits of every copy
struct RandomAccess {
    size_t bits, vec_size, num_copies, num_extra_constants;
    explicit RandomAccess(size_t b) : bits(b), vec_size(size_t(1) << b) {
        num_copies = std::min<size_t>(80 / (2 + vec_size), 135 / (2 + vec_size + bits));
        num_extra_constants = std::min<size_t>(80 - (2 + vec_size) * num_copies, 2);
    }
    size_t wire_access_index(size_t c) const { return (2 + vec_size) * c; }
    size_t wire_claimed_element(size_t c) const { return (2 + vec_size) * c + 1; }
    size_t wire_list_item(size_t i, size_t c) const { return (2 + vec_size) * c + 2 + i; }
    size_t wire_extra_constant(size_t i) const { return (2 + vec_size) * num_copies + i; }
    size_t num_routed_wires() const { return (2 + vec_size) * num_copies + num_extra_constants; }
    size_t wire_bit(size_t i, size_t c) const { return num_routed_wires() + c * bits + i; }
    size_t num_constraints() const { return num_copies * (bits + 2) + num_extra_constants; }
};
static const size_t EXP_POWER_BITS = 66;                 // wires: 0 base, 1..66 power bits (little-endian), 67 output, 68..133 intermediate values
static const size_t LOOKUP_SLOTS = 40, LOOKUP_TABLE_SLOTS = 26;      // gates/lookup.rs:41-44 (routed / 2), gates/lookup_table.rs:47-50 (routed / 3)
static const size_t NUM_COINS_LOOKUP = 4;                           // circuit_builder.rs:56-58: ChallengeA, ChallengeB, ChallengeAlpha, ChallengeDelta
enum { LU_CH_A = 0, LU_CH_B = 1, LU_CH_ALPHA = 2, LU_CH_DELTA = 3 };
enum { LU_SEL_TRANS_SRE = 0, LU_SEL_TRANS_LDC = 1, LU_SEL_INIT_SRE = 2, LU_SEL_LAST_LDC = 3, LU_SEL_START_END = 4 };      // gates/selectors.rs:34-40
static const size_t BASE_SUM_LIMBS = 63;                 // gates/base_sum.rs:31-35: min(log_floor(p - 1, 2) = 63, num_routed_wires - 1 = 79)
static const size_t UNUSED_SELECTOR = 0xFFFFFFFFull;     // selectors.rs:14

static inline unsigned gate_degree(GateType g, size_t param = 0) {
    if (g == GATE_RANDOM_ACCESS) return (unsigned)param + 1;      // random_access.rs:281-283
    switch (g) { case GATE_NOOP: return 0; case GATE_CONSTANT: return 1; case GATE_PUBLIC_INPUT: return 1;
                 case GATE_ARITHMETIC: return 3; case GATE_BASE_SUM: return 2 /* base_sum.rs:139-141 */;
                 case GATE_LOOKUP: case GATE_LOOKUP_TABLE: return 0 /* lookup.rs:131-133, lookup_table.rs:150-152 */;
                 case GATE_EXPONENTIATION: return 4 /* exponentiation.rs:186-188 */; default: return 7; }
}
static inline std::string gate_id(GateType g, size_t param = 0) {           // Gate::id(): the sort key next to the degree
    switch (g) {
        case GATE_RANDOM_ACCESS: {         // format!("{self:?}<D={D}>") (random_access.rs:119-121)
            RandomAccess ra(param);
            return "RandomAccessGate { bits: " + std::to_string(ra.bits) + ", num_copies: " + std::to_string(ra.num_copies) + ", num_extra_constants: " +
                   std::to_string(ra.num_extra_constants) + ", _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>";
        }
        case GATE_NOOP: return "NoopGate";
        case GATE_CONSTANT: return "ConstantGate { num_consts: 2 }";
        case GATE_PUBLIC_INPUT: return "PublicInputGate";
        case GATE_ARITHMETIC: return "ArithmeticGate { num_ops: 20 }";
        case GATE_BASE_SUM: return "BaseSumGate { num_limbs: 63 } + Base: 2";          // base_sum.rs:49-51
        // format!("{self:?}") of the gate structs (lookup.rs:55-57, lookup_table.rs:66-68) up to the table: finish_build() appends the
        // table's text (and last_lut_row), which orders the gates of several tables ("LookupGate {" < "LookupTableGate {" < "NoopGate")
        case GATE_LOOKUP: return "LookupGate { num_slots: 40, lut: [";
        case GATE_LOOKUP_TABLE: return "LookupTableGate { num_slots: 26, lut: [";
        case GATE_EXPONENTIATION:          // format!("{self:?}<D={D}>") (exponentiation.rs:75-77)
            return "ExponentiationGate { num_power_bits: 66, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>";
        default: return "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>";
    }
}
static inline size_t gate_num_constraints(GateType g, size_t param = 0) {
    if (g == GATE_RANDOM_ACCESS) return RandomAccess(param).num_constraints();      // random_access.rs:285-288
    switch (g) { case GATE_NOOP: return 0; case GATE_CONSTANT: return 2; case GATE_PUBLIC_INPUT: return 4;
                 case GATE_ARITHMETIC: return 20; case GATE_BASE_SUM: return 1 + BASE_SUM_LIMBS /* base_sum.rs:144-146 */;
                 case GATE_LOOKUP: case GATE_LOOKUP_TABLE: return 0;          // "No main trace constraints for lookups" (lookup.rs:72-75)
                 case GATE_EXPONENTIATION: return EXP_POWER_BITS + 1;          // exponentiation.rs:190-192
                 default: return 123; }   // poseidon.rs:403-409
}

// PoseidonGate wire layout (gates/poseidon.rs:36-96)
struct PoseidonWires {
    static constexpr int INPUT = 0, OUTPUT = 12, SWAP = 24, DELTA = 25, FULL0 = 29, PARTIAL = 29 + 12 * 3,
                         FULL1 = PARTIAL + 22, END = FULL1 + 12 * 4;   // END = 135
    static int full_sbox_0(int round, int i) { return FULL0 + 12 * (round - 1) + i; }
    static int partial_sbox(int round) { return PARTIAL + round; }
    static int full_sbox_1(int round, int i) { return FULL1 + 12 * round + i; }
};

// ---- Poseidon layers over K (the *_field variants of hash/poseidon.rs) -----------------------------
template <class K> static inline void pk_constant_layer(K (&s)[12], int round) {
    for (int i = 0; i < 12; i++) s[i] = kadd(s[i], kconst<K>(POSEIDON_RC[12 * round + i]));
}
template <class K> static inline void pk_mds_layer(K (&s)[12]) {
    K out[12];
    for (int r = 0; r < 12; r++) {
        K acc = kconst<K>(0);
        for (int i = 0; i < 12; i++) acc = kadd(acc, kscal(s[(i + r) % 12], POSEIDON_MDS_CIRC[i]));
        acc = kadd(acc, kscal(s[r], POSEIDON_MDS_DIAG[r]));
        out[r] = acc;
    }
    for (int i = 0; i < 12; i++) s[i] = out[i];
}
template <class K> static inline void pk_partial_init(K (&s)[12]) {
    for (int i = 0; i < 12; i++) s[i] = kadd(s[i], kconst<K>(POSEIDON_PARTIAL_FIRST_RC[i]));
    K t[12]; t[0] = s[0];
    for (int c = 1; c < 12; c++) {
        K acc = kconst<K>(0);
        for (int r = 1; r < 12; r++) acc = kadd(acc, kscal(s[r], POSEIDON_PARTIAL_INIT[(r - 1) * 11 + (c - 1)]));
        t[c] = acc;
    }
    for (int i = 0; i < 12; i++) s[i] = t[i];
}
template <class K> static inline void pk_partial_fast(K (&s)[12], int r) {
    K d = kscal(s[0], POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]);
    for (int i = 1; i < 12; i++) d = kadd(d, kscal(s[i], POSEIDON_PARTIAL_ROW[r * 11 + i - 1]));
    for (int i = 1; i < 12; i++) s[i] = kadd(s[i], kscal(s[0], POSEIDON_PARTIAL_COL[r * 11 + i - 1]));
    s[0] = d;
}

// ---- unfiltered constraints ------------------------------------------------------------------------
// `consts` = the gate's own constants (local_constants after the selector prefix is removed), `w` = 135 wires
template <class K>
static inline void eval_arithmetic(const K* consts, const K* w, K* out) {       // arithmetic_base.rs:72-92
    for (int i = 0; i < 20; i++) {
        K computed = kadd(kmul(kmul(w[4 * i], w[4 * i + 1]), consts[0]), kmul(w[4 * i + 2], consts[1]));
        out[i] = ksub(w[4 * i + 3], computed);
    }
}
template <class K>
static inline void eval_constant(const K* consts, const K* w, K* out) {         // constant.rs:59-66
    for (int i = 0; i < 2; i++) out[i] = ksub(consts[i], w[i]);
}
template <class K>
static inline void eval_public_input(const u64* pi_hash, const K* w, K* out) {  // public_input.rs:44-49
    for (int i = 0; i < 4; i++) out[i] = ksub(w[i], kconst<K>(pi_hash[i]));
}
template <class K>
static inline void eval_base_sum(const K* w, K* out) {                          // base_sum.rs:63-76 / 153-170, B = 2
    K computed = kconst<K>(0);                                                  // reduce_with_powers(limbs, 2) (plonk_common.rs:116-128)
    for (size_t i = BASE_SUM_LIMBS; i-- > 0;) computed = kadd(kadd(computed, computed), w[1 + i]);
    out[0] = ksub(computed, w[0]);
    for (size_t i = 0; i < BASE_SUM_LIMBS; i++) out[1 + i] = kmul(w[1 + i], ksub(w[1 + i], kconst<K>(1)));      // (limb - 0)(limb - 1)
}
template <class K>
static inline void eval_random_access(size_t bits, const K* consts, const K* w, K* out) {      // random_access.rs:139-184
    const RandomAccess ra(bits);
    size_t c = 0;
    for (size_t copy = 0; copy < ra.num_copies; copy++) {
        std::vector<K> items(ra.vec_size);
        for (size_t i = 0; i < ra.vec_size; i++) items[i] = w[ra.wire_list_item(i, copy)];
        for (size_t i = 0; i < bits; i++) { const K b = w[ra.wire_bit(i, copy)]; out[c++] = kmul(b, ksub(b, kconst<K>(1))); }
        K rec = kconst<K>(0);
        for (size_t i = bits; i-- > 0;) rec = kadd(kadd(rec, rec), w[ra.wire_bit(i, copy)]);
        out[c++] = ksub(rec, w[ra.wire_access_index(copy)]);
        for (size_t i = 0; i < bits; i++) {            // fold the list: the left or right item of each pair by bit i
            const K b = w[ra.wire_bit(i, copy)];
            for (size_t j = 0; 2 * j + 1 < items.size(); j++) items[j] = kadd(items[2 * j], kmul(b, ksub(items[2 * j + 1], items[2 * j])));
            items.resize(items.size() / 2);
        }
        out[c++] = ksub(items[0], w[ra.wire_claimed_element(copy)]);
    }
    for (size_t i = 0; i < ra.num_extra_constants; i++) out[c++] = ksub(consts[i], w[ra.wire_extra_constant(i)]);
}
template <class K>
static inline void eval_exponentiation(const K* w, K* out) {                   // exponentiation.rs:88-124: square-and-multiply, bits big-endian
    const size_t n = EXP_POWER_BITS;
    const K base = w[0], one = kconst<K>(1);
    for (size_t i = 0; i < n; i++) {
        const K prev = i == 0 ? one : kmul(w[2 + n + i - 1], w[2 + n + i - 1]);
        const K bit = w[1 + (n - 1 - i)];
        out[i] = ksub(kmul(prev, kadd(kmul(bit, base), ksub(one, bit))), w[2 + n + i]);
    }
    out[n] = ksub(w[1 + n], w[2 + n + n - 1]);
}
template <class K>
static inline void eval_poseidon(const K* w, K* out) {                          // poseidon.rs:113-191
    typedef PoseidonWires PW;
    int c = 0;
    K swap = w[PW::SWAP];
    out[c++] = kmul(swap, ksub(swap, kconst<K>(1)));
    for (int i = 0; i < 4; i++) out[c++] = ksub(kmul(swap, ksub(w[PW::INPUT + i + 4], w[PW::INPUT + i])), w[PW::DELTA + i]);
    K s[12];
    for (int i = 0; i < 4; i++) {
        s[i] = kadd(w[PW::INPUT + i], w[PW::DELTA + i]);
        s[i + 4] = ksub(w[PW::INPUT + i + 4], w[PW::DELTA + i]);
    }
    for (int i = 8; i < 12; i++) s[i] = w[PW::INPUT + i];
    int round = 0;
    for (int r = 0; r < 4; r++) {
        pk_constant_layer(s, round);
        if (r != 0)
            for (int i = 0; i < 12; i++) { K in = w[PW::full_sbox_0(r, i)]; out[c++] = ksub(s[i], in); s[i] = in; }
        for (int i = 0; i < 12; i++) s[i] = ksbox(s[i]);
        pk_mds_layer(s);
        round++;
    }
    pk_partial_init(s);
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        K in = w[PW::partial_sbox(r)];
        out[c++] = ksub(s[0], in);
        s[0] = kadd(ksbox(in), kconst<K>(POSEIDON_PARTIAL_RC[r]));   // last constant is 0 (poseidon.rs:186-189)
        pk_partial_fast(s, r);
    }
    round += POSEIDON_PARTIAL_ROUNDS;
    for (int r = 0; r < 4; r++) {
        pk_constant_layer(s, round);
        for (int i = 0; i < 12; i++) { K in = w[PW::full_sbox_1(r, i)]; out[c++] = ksub(s[i], in); s[i] = in; }
        for (int i = 0; i < 12; i++) s[i] = ksbox(s[i]);
        pk_mds_layer(s);
        round++;
    }
    for (int i = 0; i < 12; i++) out[c++] = ksub(s[i], w[PW::OUTPUT + i]);
    assert(c == 123);
}

// Selector bookkeeping (selectors.rs:110-185)
struct SelectorsInfo {
    std::vector<GateType> gates;                 // sorted by (degree, id): circuit_builder.rs:984-986
    std::vector<size_t> gate_params;               // per gate: the lookup table of a LookupGate / LookupTableGate, the bits of a RandomAccessGate (else 0)
    std::vector<size_t> selector_indices;        // per gate: which selector polynomial
    std::vector<std::pair<size_t, size_t>> groups;   // [start, end) ranges of gate indices
    size_t num_selectors() const { return groups.size(); }
    size_t num_lookup_selectors = 0;             // lookup selector columns between the gate selectors and the gates' constants
    size_t gate_index(GateType g, size_t lut = 0) const {
        for (size_t i = 0; i < gates.size(); i++) if (gates[i] == g && (gate_params.empty() || gate_params[i] == lut)) return i;
        assert(false); return 0;
    }
};

template <class K>
static inline K compute_filter(size_t row, std::pair<size_t, size_t> group, K s, bool many) {   // gate.rs:277-284
    K f = kconst<K>(1);
    for (size_t i = group.first; i < group.second; i++)
        if (i != row) f = kmul(f, ksub(kconst<K>(i), s));
    if (many) f = kmul(f, ksub(kconst<K>(UNUSED_SELECTOR), s));
    return f;
}

// Sum of all gates' filtered constraints, slot-wise (vanishing_poly.rs:671-699 / 706-732)
template <class K>
static inline void evaluate_gate_constraints(const SelectorsInfo& si, size_t num_gate_constraints,
                                             const K* local_constants, const K* wires, const u64* pi_hash, K* out) {
    for (size_t i = 0; i < num_gate_constraints; i++) out[i] = kconst<K>(0);
    const size_t nsel = si.num_selectors();
    const K* gc = local_constants + nsel + si.num_lookup_selectors;       // vars.remove_prefix(num_selectors + num_lookup_selectors) (gate.rs:129-133)
    K tmp[123];
    for (size_t gi = 0; gi < si.gates.size(); gi++) {
        const size_t sel = si.selector_indices[gi];
        K filter = compute_filter<K>(gi, si.groups[sel], local_constants[sel], nsel > 1);
        const size_t param = si.gate_params.empty() ? 0 : si.gate_params[gi];
        size_t nc = gate_num_constraints(si.gates[gi], param);
        switch (si.gates[gi]) {
            case GATE_RANDOM_ACCESS: eval_random_access<K>(param, gc, wires, tmp); break;
            case GATE_NOOP: case GATE_LOOKUP: case GATE_LOOKUP_TABLE: break;
            case GATE_CONSTANT: eval_constant<K>(gc, wires, tmp); break;
            case GATE_PUBLIC_INPUT: eval_public_input<K>(pi_hash, wires, tmp); break;
            case GATE_ARITHMETIC: eval_arithmetic<K>(gc, wires, tmp); break;
            case GATE_BASE_SUM: eval_base_sum<K>(wires, tmp); break;
            case GATE_EXPONENTIATION: eval_exponentiation<K>(wires, tmp); break;
            default: eval_poseidon<K>(wires, tmp); break;
        }
        for (size_t j = 0; j < nc; j++) out[j] = kadd(out[j], kmul(filter, tmp[j]));
    }
}

}  // namespace orc
