// ORACLE (test infrastructure only).  CPU restatement of circuit construction + witness generation for the
// matrix-multiplication demo circuit C = A*B over Goldilocks.
// Follows:
//   plonky2/src/bin/matrix_mul.rs:25-81, plonky2/examples/matrix_multiplication.rs:21-67 (the circuit)
//   plonky2/src/gadgets/arithmetic.rs:34-159,187-213      (arithmetic(), special cases, add, mul)
//   plonky2/src/plonk/circuit_builder.rs:353-388,424-436,485-496,665-695 (add_gate, connect, constant, find_slot)
//   plonky2/src/plonk/circuit_builder.rs:913-1146         (build(): PI hashing, PublicInputGate, ConstantGate,
//                                                         padding, selectors, sigmas, commitment, digest)
//   plonky2/src/hash/hashing.rs:31-59, hash/poseidon.rs:724-751 (in-circuit sponge, PoseidonGate routing)
//   plonky2/src/plonk/permutation_argument.rs:21-170      (copy-constraint partition -> sigma)
//   plonky2/src/gates/selectors.rs:110-185, field/src/cosets.rs:9-24
//   plonky2/src/iop/witness.rs:340-353, iop/generator.rs:19-98,286-304, gates/poseidon.rs:411-497 (witness)
//   plonky2/src/plonk/circuit_data.rs:72-90               (standard_recursion_config)
#pragma once
#include "gl_batch.hpp"
#include "gl_gates.hpp"
#include <algorithm>
#include <map>
#include <numeric>

namespace orc {

struct CircuitConfig {                       // circuit_data.rs:72-90
    size_t num_wires = 135, num_routed_wires = 80, num_constants = 2, num_challenges = 2;
    size_t max_quotient_degree_factor = 8;
    unsigned rate_bits = 3, cap_height = 4, proof_of_work_bits = 16, num_query_rounds = 28;
    unsigned fri_arity_bits = 4, fri_final_poly_bits = 5;      // ConstantArityBits(4, 5)
};

struct Target {
    bool is_wire; size_t row, col;           // wire
    size_t index;                            // virtual
    static Target wire(size_t r, size_t c) { return Target{true, r, c, 0}; }
    static Target virt(size_t i) { return Target{false, 0, 0, i}; }
    bool operator==(const Target& o) const { return is_wire == o.is_wire && row == o.row && col == o.col && index == o.index; }
};

struct ArithOp { size_t row, slot; u64 c0, c1; };

struct CommonData {
    CircuitConfig config;
    unsigned degree_bits = 0;
    SelectorsInfo selectors;
    size_t num_gate_constraints = 0, num_constants = 0, num_public_inputs = 0, num_partial_products = 0;
    size_t quotient_degree_factor = 8;
    size_t num_lookup_polys = 0, num_lookup_selectors = 0;      // circuit_data.rs:376-381: per challenge 1 RE + ceil(40 / 7) partial SLDC polynomials
    std::vector<size_t> last_lut_rows;                          // per table: LookupTableGate's third field (lookup_table.rs:31-32), its lowest row
    std::vector<std::vector<std::pair<uint16_t, uint16_t>>> luts;   // the lookup tables: (input, output) pairs
    std::vector<u64> k_is;
    std::vector<unsigned> fri_reduction_arity_bits;
    size_t degree() const { return size_t(1) << degree_bits; }
    size_t final_poly_len() const { unsigned t = 0; for (auto a : fri_reduction_arity_bits) t += a; return size_t(1) << (degree_bits - t); }
};

struct CircuitData {
    CommonData common;
    size_t m = 0;
    // prover-only
    std::vector<GateType> row_gate;                     // gate type of each row
    std::vector<std::vector<u64>> constants_sigmas;     // column-major values: 4 constants + 80 sigmas, each n
    std::vector<u64> subgroup;
    PolynomialBatch constants_sigmas_commitment;
    Digest circuit_digest;
    // witness recipe
    size_t num_targets = 0;
    std::vector<size_t> representative;                 // union-find result over all targets (wires then virtual)
    std::vector<Target> a_targets, b_targets, public_inputs;
    std::vector<ArithOp> arith_ops;                     // in generator (creation) order
    std::vector<size_t> poseidon_rows;
    struct LookupWire { size_t last_lu_gate, last_lut_gate, first_lut_gate; };      // circuit_builder.rs:73-85 (the gate rows are upside down)
    std::vector<LookupWire> lookup_rows;
    std::vector<std::vector<std::pair<Target, Target>>> lut_to_lookups;             // per table: (looking_in, looking_out) targets
    struct LookupSlot { size_t row, slot, lut; };
    std::vector<LookupSlot> lookup_slots;
    struct SplitRecipe { Target integer; std::vector<size_t> rows; };
    std::vector<SplitRecipe> split_ops;                 // BaseSumGate rows: sums from the integer, limbs from the sums
    std::vector<size_t> exp_rows;                       // ExponentiationGate rows (one ExponentiationGenerator each)
    struct RaRecipe { size_t row, copy, bits; };
    std::vector<RaRecipe> ra_ops;                       // used RandomAccessGate copies (one RandomAccessGenerator each)
    size_t pi_row = 0;
    struct ConstantWire { size_t row, wire; u64 value; };
    std::vector<ConstantWire> constant_wires;           // ConstantGenerator outputs: (row, wire column, value), one ConstantGate per two constants
    size_t target_index(const Target& t) const {        // iop/target.rs: wires first, then virtual targets
        return t.is_wire ? t.row * common.config.num_wires + t.col : common.degree() * common.config.num_wires + t.index;
    }
};

// ---- the builder -------------------------------------------------------------------------------------
struct CircuitBuilder {
    CircuitConfig config;
    struct GateInstance { GateType type; u64 constants[2]; size_t lut; };       // lut: the table of a LookupGate / LookupTableGate
    std::vector<GateInstance> gate_instances;
    std::vector<Target> public_inputs;
    size_t virtual_target_index = 0;
    std::vector<std::pair<Target, Target>> copy_constraints;
    std::map<u64, Target> constants_to_targets;                 // keyed by canonical value
    std::map<size_t, u64> virtual_constants;                    // virtual index -> constant (targets_to_constants)
    std::map<std::pair<u64, u64>, std::pair<size_t, size_t>> current_slots;   // ArithmeticGate params -> (row, slot)
    struct ConstGen { size_t row, constant_index, wire_index; };
    std::vector<ConstGen> constant_generators;
    std::vector<ArithOp> arith_ops;
    std::vector<size_t> poseidon_rows;
    typedef std::tuple<bool, size_t, size_t, size_t> TargetKey;
    std::map<std::tuple<u64, u64, TargetKey, TargetKey, TargetKey>, Target> arithmetic_results;

    Target add_virtual_target() { return Target::virt(virtual_target_index++); }
    size_t add_gate(GateType t, u64 c0 = 0, u64 c1 = 0, size_t lut = 0) {        // circuit_builder.rs:353-388
        size_t row = gate_instances.size();
        if (t == GATE_CONSTANT) for (size_t i = 0; i < config.num_constants; i++) constant_generators.push_back({row, i, i});
        if (t == GATE_RANDOM_ACCESS) {                            // extra_constant_wires (random_access.rs:290-294; circuit_builder.rs:364-372)
            const RandomAccess ra(lut);
            for (size_t i = 0; i < ra.num_extra_constants; i++) constant_generators.push_back({row, i, ra.wire_extra_constant(i)});
        }
        gate_instances.push_back({t, {c0, c1}, lut});
        return row;
    }
    bool record_copies = true;          // false for a verifier-only build (CommonData without sigmas / commitment)
    void connect(Target x, Target y) { if (record_copies) copy_constraints.push_back({x, y}); }      // :424-436
    Target constant(u64 c) {                                                      // :485-496
        c = canon(c);
        auto it = constants_to_targets.find(c);
        if (it != constants_to_targets.end()) return it->second;
        Target t = add_virtual_target();
        constants_to_targets[c] = t;
        virtual_constants[t.index] = c;
        return t;
    }
    Target zero() { return constant(0); }
    Target one() { return constant(1); }
    bool target_as_constant(const Target& t, u64* out) const {
        if (t.is_wire) return false;
        auto it = virtual_constants.find(t.index);
        if (it == virtual_constants.end()) return false;
        *out = it->second; return true;
    }
    // gadgets/arithmetic.rs:34-159 (base arithmetic gate path; the memo table cannot hit for this circuit)
    Target arithmetic(u64 c0, u64 c1, Target m0, Target m1, Target addend) {
        Target z = zero();
        u64 v0 = 0, v1 = 0, va = 0;
        bool k0 = target_as_constant(m0, &v0), k1 = target_as_constant(m1, &v1), ka = target_as_constant(addend, &va);
        bool first_zero = c0 == 0 || m0 == z || m1 == z;
        bool second_zero = c1 == 0 || addend == z;
        bool first_known = first_zero || (k0 && k1), second_known = second_zero || ka;
        if (first_known && second_known) {
            u64 x = first_zero ? 0 : mul(mul(v0, v1), c0), y = second_zero ? 0 : mul(va, c1);
            return constant(add(x, y));
        }
        if (first_zero && c1 == 1) return addend;
        if (second_zero) {
            if (k0 && canon(mul(v0, c0)) == 1) return m1;
            if (k1 && canon(mul(v1, c0)) == 1) return m0;
        }
        // base_arithmetic_results (arithmetic.rs:64-75): the same operation on the same targets is not placed twice
        auto tkey = [](const Target& t) { return std::make_tuple(t.is_wire, t.row, t.col, t.index); };
        auto memo_key = std::make_tuple(canon(c0), canon(c1), tkey(m0), tkey(m1), tkey(addend));
        auto hit = arithmetic_results.find(memo_key);
        if (hit != arithmetic_results.end()) return hit->second;
        // find_slot(ArithmeticGate, params = constants = [c0, c1])  (circuit_builder.rs:665-695)
        auto key = std::make_pair(c0, c1);
        size_t row, slot;
        auto it = current_slots.find(key);
        if (it != current_slots.end()) { row = it->second.first; slot = it->second.second; }
        else { row = add_gate(GATE_ARITHMETIC, c0, c1); slot = 0; }
        if (slot == 20 - 1) current_slots.erase(key); else current_slots[key] = {row, slot + 1};
        connect(m0, Target::wire(row, 4 * slot));
        connect(m1, Target::wire(row, 4 * slot + 1));
        connect(addend, Target::wire(row, 4 * slot + 2));
        arith_ops.push_back({row, slot, c0, c1});
        arithmetic_results[memo_key] = Target::wire(row, 4 * slot + 3);
        return Target::wire(row, 4 * slot + 3);
    }
    Target mul_t(Target x, Target y) { return arithmetic(1, 0, x, y, x); }                 // arithmetic.rs:210-213
    Target add_t(Target x, Target y) { Target o = one(); return arithmetic(1, 1, x, o, y); }   // :187-191
    Target mul_const(u64 c, Target x) { Target ct = constant(c); return mul_t(ct, x); }      // :169-172
    Target add_const(Target x, u64 c) { Target ct = constant(c); return add_t(x, ct); }      // :163-166

    // ---- lookups (gadgets/lookup.rs:56-125, circuit_builder.rs:222-252,609-660) ----
    std::vector<std::vector<std::pair<uint16_t, uint16_t>>> luts;
    std::vector<std::vector<std::pair<Target, Target>>> lut_to_lookups;
    struct LookupWire { size_t last_lu_gate, last_lut_gate, first_lut_gate; };
    std::vector<LookupWire> lookup_rows;
    struct LookupSlot { size_t row, slot, lut; };
    std::vector<LookupSlot> lookup_slots;                                        // one LookupGenerator each (lookup.rs:105-118)
    std::map<std::pair<int, u64>, std::pair<size_t, size_t>> gate_slots;      // find_slot: (gate type, parameter) -> (row, next slot)
    size_t add_lookup_table_from_pairs(const std::vector<std::pair<uint16_t, uint16_t>>& table) {
        for (size_t i = 0; i < luts.size(); i++) if (luts[i] == table) return i;      // is_stored
        luts.push_back(table); lut_to_lookups.emplace_back();
        return luts.size() - 1;
    }
    Target add_lookup_from_index(Target looking_in, size_t lut_index) {
        assert(lut_index < luts.size());
        Target looking_out = add_virtual_target();
        lut_to_lookups[lut_index].push_back({looking_in, looking_out});
        return looking_out;
    }
    std::pair<size_t, size_t> find_slot(GateType t, u64 param, size_t num_ops) {  // circuit_builder.rs:665-695 (no constants for these gates)
        auto key = std::make_pair((int)t, param);
        auto it = gate_slots.find(key);
        size_t row, slot;
        if (it != gate_slots.end()) { row = it->second.first; slot = it->second.second; }
        else { row = add_gate(t, 0, 0, (t == GATE_LOOKUP || t == GATE_LOOKUP_TABLE || t == GATE_RANDOM_ACCESS) ? (size_t)param : 0); slot = 0; }
        if (slot == num_ops - 1) gate_slots.erase(key); else gate_slots[key] = {row, slot + 1};
        return {row, slot};
    }
    void add_all_lookups() {                                                      // gadgets/lookup.rs:79-125
        for (size_t li = 0; li < luts.size(); li++) {
            assert(!lut_to_lookups[li].empty() && "LUT is unused");
            const size_t last_lu_gate = gate_instances.size();
            for (auto& lk : lut_to_lookups[li]) {
                auto rs = find_slot(GATE_LOOKUP, (u64)li, LOOKUP_SLOTS);
                lookup_slots.push_back({rs.first, rs.second, li});
                connect(Target::wire(rs.first, 2 * rs.second), lk.first);
                connect(Target::wire(rs.first, 2 * rs.second + 1), lk.second);
            }
            const size_t last_lut_gate = gate_instances.size();
            const size_t num_lut_rows = (luts[li].size() - 1) / LOOKUP_TABLE_SLOTS + 1;
            for (size_t c = 0; c < LOOKUP_TABLE_SLOTS * num_lut_rows; c++) (void)find_slot(GATE_LOOKUP_TABLE, (u64)li, LOOKUP_TABLE_SLOTS);
            const size_t first_lut_gate = gate_instances.size() - 1;
            add_gate(GATE_NOOP);          // the row after the table is all zeros: initial constraints become a zero check
            lookup_rows.push_back({last_lu_gate, last_lut_gate, first_lut_gate});
        }
    }

    // split_le (gadgets/split_join.rs:19-63) / range_check (gadgets/range_check.rs:14-16): k BaseSumGate<2> rows of 63 limbs,
    // unused bits tied to zero, the gates' sums recombined with mul_const_add(2^63, acc, sum) and tied to the integer
    struct SplitOp { Target integer; std::vector<size_t> rows; };
    std::vector<SplitOp> split_ops;                               // WireSplitGenerator + one BaseSplitGenerator per row
    std::vector<Target> split_le(Target integer, size_t num_bits) {
        std::vector<Target> bits;
        if (num_bits == 0) return bits;
        const size_t k = (num_bits + BASE_SUM_LIMBS - 1) / BASE_SUM_LIMBS;
        std::vector<size_t> rows;
        for (size_t i = 0; i < k; i++) rows.push_back(add_gate(GATE_BASE_SUM));
        for (size_t row : rows) for (size_t c = 1; c <= BASE_SUM_LIMBS; c++) bits.push_back(Target::wire(row, c));
        for (size_t i = num_bits; i < bits.size(); i++) connect(bits[i], zero());      // assert_zero (circuit_builder.rs)
        bits.resize(num_bits);
        Target z = zero(), acc = z;
        const u64 base = u64(1) << BASE_SUM_LIMBS;                                     // F::TWO.exp_u64(num_limbs)
        for (size_t i = rows.size(); i-- > 0;) {
            Target sum = Target::wire(rows[i], 0);
            Target bt = constant(base);
            acc = arithmetic(1, 1, bt, acc, sum);                                      // mul_const_add(base, acc, sum) = mul_add(c, acc, sum)
        }
        connect(acc, integer);
        split_ops.push_back({integer, rows});
        return bits;
    }
    void range_check(Target x, size_t n_log) { (void)split_le(x, n_log); }

    // random_access (gadgets/random_access.rs:14-47): one copy of a RandomAccessGate(bits = log2 |v|); its RandomAccessGenerator
    // (random_access.rs:335-390) sets the claimed element and the index bits
    struct RaOp { size_t row, copy, bits; };
    std::vector<RaOp> ra_ops;
    Target random_access(Target access_index, const std::vector<Target>& v) {
        if (v.size() == 1) return v[0];
        size_t bits = 0; while ((size_t(1) << bits) < v.size()) bits++;
        assert((size_t(1) << bits) == v.size() && bits <= 6);
        const RandomAccess ra(bits);
        Target claimed = add_virtual_target();
        auto rs = find_slot(GATE_RANDOM_ACCESS, (u64)bits, ra.num_copies);
        for (size_t i = 0; i < v.size(); i++) connect(v[i], Target::wire(rs.first, ra.wire_list_item(i, rs.second)));
        connect(access_index, Target::wire(rs.first, ra.wire_access_index(rs.second)));
        connect(claimed, Target::wire(rs.first, ra.wire_claimed_element(rs.second)));
        ra_ops.push_back({rs.first, rs.second, bits});
        return claimed;
    }
    // le_sum (gadgets/split_base.rs:37-80) for the bit counts whose arithmetic form is the cheaper one (num_bits - 1 <= 20 operations)
    Target le_sum(const std::vector<Target>& bits) {
        if (bits.empty()) return zero();
        assert(bits.size() - 1 <= 20);
        Target two = constant(2), sum = bits.back();
        for (size_t i = bits.size() - 1; i-- > 0;) sum = arithmetic(1, 1, two, sum, bits[i]);      // mul_add(two, sum, bit)
        return sum;
    }

    // exp_from_bits / exp (gadgets/arithmetic.rs:240-272): ONE ExponentiationGate row of 66 power bits, the bits beyond the exponent's
    // tied to _false(); the gate's ExponentiationGenerator (exponentiation.rs:233-280) fills the intermediate values and the output
    std::vector<size_t> exp_rows;
    Target exp_from_bits(Target base, const std::vector<Target>& exponent_bits) {
        assert(exponent_bits.size() <= EXP_POWER_BITS);
        Target f = zero();                                        // self._false()
        const size_t row = add_gate(GATE_EXPONENTIATION);
        connect(base, Target::wire(row, 0));
        for (size_t i = 0; i < EXP_POWER_BITS; i++) connect(i < exponent_bits.size() ? exponent_bits[i] : f, Target::wire(row, 1 + i));
        exp_rows.push_back(row);
        return Target::wire(row, 1 + EXP_POWER_BITS);
    }
    Target exp(Target base, Target exponent, size_t num_bits) { return exp_from_bits(base, split_le(exponent, num_bits)); }

    // permute_swapped (hash/poseidon.rs:724-751): one PoseidonGate row; `swap` exchanges the first two digests of the input
    std::array<Target, 12> permute_swapped(const std::array<Target, 12>& inputs, Target swap) {
        size_t row = add_gate(GATE_POSEIDON);
        poseidon_rows.push_back(row);
        connect(swap, Target::wire(row, PoseidonWires::SWAP));
        for (int i = 0; i < 12; i++) connect(inputs[i], Target::wire(row, PoseidonWires::INPUT + i));
        std::array<Target, 12> out;
        for (int i = 0; i < 12; i++) out[i] = Target::wire(row, PoseidonWires::OUTPUT + i);
        return out;
    }
    // hash_or_noop (hash/hashing.rs:15-22)
    std::array<Target, 4> hash_or_noop(const std::vector<Target>& inputs) {
        if (inputs.size() <= 4) { Target z = zero(); std::array<Target, 4> h; h.fill(z); for (size_t i = 0; i < inputs.size(); i++) h[i] = inputs[i]; return h; }
        return hash_public_inputs(inputs);        // hash_n_to_hash_no_pad: the same sponge
    }
    // verify_merkle_proof (hash/merkle_proofs.rs:78-150) against a ROOT (a cap of one digest: cap_index = le_sum of no bits = zero and
    // random_access over a one-element list is the element itself, gadgets/random_access.rs:17-20): the leaf hash, one swapped permutation
    // per level on (state || sibling || zeros), the result tied to the root
    void verify_merkle_proof(const std::vector<Target>& leaf_data, const std::vector<Target>& leaf_index_bits, const std::array<Target, 4>& root,
                             const std::vector<std::array<Target, 4>>& siblings) {
        Target z = zero();
        std::array<Target, 4> state = hash_or_noop(leaf_data);
        for (size_t l = 0; l < siblings.size(); l++) {
            std::array<Target, 12> in; in.fill(z);
            for (int i = 0; i < 4; i++) { in[i] = state[i]; in[4 + i] = siblings[l][i]; }
            auto out = permute_swapped(in, leaf_index_bits[l]);
            for (int i = 0; i < 4; i++) state[i] = out[i];
        }
        for (int i = 0; i < 4; i++) connect(root[i], state[i]);
    }

    // hash_n_to_hash_no_pad in circuit (hashing.rs:24-59) with PoseidonGate routing (poseidon.rs:724-751)
    std::array<Target, 4> hash_public_inputs(const std::vector<Target>& inputs) {
        Target z = zero();
        std::array<Target, 12> state; state.fill(z);
        for (size_t off = 0; off < inputs.size(); off += 8) {
            size_t c = std::min<size_t>(8, inputs.size() - off);
            for (size_t i = 0; i < c; i++) state[i] = inputs[off + i];
            Target f = zero();                                  // self._false()
            size_t row = add_gate(GATE_POSEIDON);
            poseidon_rows.push_back(row);
            connect(f, Target::wire(row, PoseidonWires::SWAP));
            for (int i = 0; i < 12; i++) connect(state[i], Target::wire(row, PoseidonWires::INPUT + i));
            for (int i = 0; i < 12; i++) state[i] = Target::wire(row, PoseidonWires::OUTPUT + i);
        }
        return {state[0], state[1], state[2], state[3]};
    }
};

static inline std::vector<unsigned> fri_reduction_arity_bits(const CircuitConfig& c, unsigned degree_bits) {
    std::vector<unsigned> r;                                     // reduction_strategies.rs:39-49
    while (degree_bits > c.fri_final_poly_bits && degree_bits + c.rate_bits - c.fri_arity_bits >= c.cap_height) {
        r.push_back(c.fri_arity_bits); degree_bits -= c.fri_arity_bits;
    }
    return r;
}

// build() (circuit_builder.rs:913-1146) for whatever the builder holds: public-input hashing, PublicInputGate, ConstantGate,
// padding, selectors, constants, sigma polynomials, the constants||sigmas commitment and the circuit digest.
// `prover_data = false` builds only what the verifier needs (CommonData): no copy constraints, sigmas or commitment.
static inline void finish_build(CircuitBuilder& b, CircuitData& cd, unsigned threads, bool prover_data) {
    // ---- build() (circuit_builder.rs:913-1146) ----
    const CircuitConfig& cfg = b.config;
    auto pi_hash = b.hash_public_inputs(b.public_inputs);
    size_t pi_row = b.add_gate(GATE_PUBLIC_INPUT);
    for (size_t i = 0; i < 4; i++) b.connect(pi_hash[i], Target::wire(pi_row, i));
    b.add_all_lookups();                                        // circuit_builder.rs:938-939: LUT-related gates come right after the PI gate
    while (b.constants_to_targets.size() > b.constant_generators.size()) b.add_gate(GATE_CONSTANT);
    {
        size_t gi = 0;
        for (auto& kv : b.constants_to_targets) {               // std::map iterates by canonical value (:946-950)
            auto g = b.constant_generators[gi++];
            b.gate_instances[g.row].constants[g.constant_index] = kv.first;
            b.connect(Target::wire(g.row, g.wire_index), kv.second);
            cd.constant_wires.push_back({g.row, g.wire_index, kv.first});
        }
    }
    while (b.gate_instances.size() & (b.gate_instances.size() - 1)) b.add_gate(GATE_NOOP);   // blind_and_pad, zk off
    const size_t degree = b.gate_instances.size();
    const unsigned degree_bits = log2_strict(degree);
    CommonData& cm = cd.common;
    cm.config = cfg; cm.degree_bits = degree_bits;
    cm.fri_reduction_arity_bits = fri_reduction_arity_bits(cfg, degree_bits);
    cm.quotient_degree_factor = cfg.max_quotient_degree_factor;

    // gates sorted by (degree, id) (:984-986).  A LookupGate / LookupTableGate is a different gate per table: its id is the Debug string of
    // the struct, table included (lookup.rs:55-57, lookup_table.rs:66-68), so the order of several tables' gates is the text order of
    // "[(0, 7), (1, 12), ...]" and, for two LookupTableGates over equal tables, of the decimal last_lut_row
    struct Kind { GateType type; size_t lut; std::string id; };
    std::vector<Kind> kinds;
    auto lut_text = [&](size_t li) {
        std::string t = "[";
        for (size_t e = 0; e < b.luts[li].size(); e++)
            t += (e ? ", (" : "(") + std::to_string(b.luts[li][e].first) + ", " + std::to_string(b.luts[li][e].second) + ")";
        return t + "]";
    };
    for (int g = 0; g < GATE_NUM_TYPES; g++)
        for (size_t li = 0; li < 8; li++) {                       // the parameter: a table index, or the bits of a RandomAccessGate
            const bool per_table = g == GATE_LOOKUP || g == GATE_LOOKUP_TABLE || g == GATE_RANDOM_ACCESS;
            if (!per_table && li) break;
            bool used = false;
            for (auto& gi : b.gate_instances) if (gi.type == (GateType)g && (!per_table || gi.lut == li)) { used = true; break; }
            if (!used) continue;
            std::string id = gate_id((GateType)g, li);
            if (g == GATE_LOOKUP) id += lut_text(li).substr(1) + " }";              // gate_id() ends with "lut: ["
            if (g == GATE_LOOKUP_TABLE) id += lut_text(li).substr(1) + ", last_lut_row: " + std::to_string(b.lookup_rows[li].last_lut_gate) + " }";
            kinds.push_back({(GateType)g, li, id});
        }
    std::sort(kinds.begin(), kinds.end(), [](const Kind& x, const Kind& y) {
        return std::make_pair(gate_degree(x.type, x.lut), x.id) < std::make_pair(gate_degree(y.type, y.lut), y.id); });
    std::vector<GateType> gates;
    SelectorsInfo& si = cm.selectors;
    for (auto& k : kinds) { gates.push_back(k.type); si.gate_params.push_back(k.lut); }
    // selector_polynomials(gates, instances, max_degree = quotient_degree_factor + 1) (selectors.rs:110-185)
    si.gates = gates;
    const size_t max_degree = cm.quotient_degree_factor + 1, num_gates = gates.size();
    const size_t max_gate_degree = gate_degree(gates.back(), si.gate_params.back());
    if (max_gate_degree + num_gates - 1 <= max_degree) {
        si.groups.push_back({0, num_gates});
        si.selector_indices.assign(num_gates, 0);
    } else {
        assert(max_gate_degree < max_degree);
        size_t start = 0;
        while (start < num_gates) {
            size_t size = 0;
            while (start + size < num_gates && size + gate_degree(gates[start + size], si.gate_params[start + size]) < max_degree) size++;
            si.groups.push_back({start, start + size});
            start += size;
        }
        for (size_t i = 0; i < num_gates; i++)
            for (size_t g = 0; g < si.groups.size(); g++) if (i >= si.groups[g].first && i < si.groups[g].second) si.selector_indices.push_back(g);
    }
    std::vector<std::vector<u64>> constant_vecs(si.groups.size(), std::vector<u64>(degree));
    for (size_t j = 0; j < degree; j++) {
        size_t i = si.gate_index(b.gate_instances[j].type, b.gate_instances[j].lut), gr = si.selector_indices[i];
        for (size_t g = 0; g < si.groups.size(); g++) constant_vecs[g][j] = (si.groups.size() == 1 || g == gr) ? i : UNUSED_SELECTOR;
    }
    // lookup selectors (circuit_builder.rs:991-1002; gates/selectors.rs:50-103): TransSre, TransLdc, InitSre, LastLdc, then one "end"
    // selector per table, between the gate selectors and the gates' constants
    if (!b.luts.empty()) {
        std::vector<std::vector<u64>> ls(LU_SEL_START_END + b.lookup_rows.size(), std::vector<u64>(degree, 0));
        for (size_t t = 0; t < b.lookup_rows.size(); t++) {
            auto& lr = b.lookup_rows[t];
            for (size_t row = lr.last_lut_gate; row <= lr.first_lut_gate; row++) ls[LU_SEL_TRANS_SRE][row] = 1;
            for (size_t row = lr.last_lu_gate; row < lr.last_lut_gate; row++) ls[LU_SEL_TRANS_LDC][row] = 1;
            ls[LU_SEL_INIT_SRE][lr.first_lut_gate + 1] = 1;
            ls[LU_SEL_LAST_LDC][lr.last_lu_gate] = 1;
            ls[LU_SEL_START_END + t][lr.last_lut_gate] = 1;
        }
        cm.num_lookup_selectors = si.num_lookup_selectors = ls.size();
        for (auto& v : ls) constant_vecs.push_back(v);
        cm.num_lookup_polys = (LOOKUP_SLOTS + (cfg.max_quotient_degree_factor - 1) - 1) / (cfg.max_quotient_degree_factor - 1) + 1;     // :1079-1085
        cm.luts = b.luts;
        for (auto& lr : b.lookup_rows) cm.last_lut_rows.push_back(lr.last_lut_gate);
    }
    // constant_polys (:822-843): max_constants over gate types used
    size_t max_constants = 0;
    for (size_t gi = 0; gi < gates.size(); gi++) {
        const GateType g = gates[gi];
        max_constants = std::max<size_t>(max_constants, g == GATE_CONSTANT || g == GATE_ARITHMETIC ? 2 : g == GATE_RANDOM_ACCESS ? RandomAccess(si.gate_params[gi]).num_extra_constants : 0);
    }
    for (size_t c = 0; c < max_constants; c++) {
        std::vector<u64> col(degree);
        for (size_t j = 0; j < degree; j++) col[j] = b.gate_instances[j].constants[c];
        constant_vecs.push_back(col);
    }
    cm.num_constants = constant_vecs.size();

    { u64 x = 1; for (size_t i = 0; i < cfg.num_routed_wires; i++) { cm.k_is.push_back(x); x = mul(x, GL_GENERATOR); } }
    cm.num_gate_constraints = 0;
    for (auto g : gates) cm.num_gate_constraints = std::max(cm.num_gate_constraints, gate_num_constraints(g));
    cm.num_partial_products = (cfg.num_routed_wires + cm.quotient_degree_factor - 1) / cm.quotient_degree_factor - 1;   // partial_products.rs:40-47
    cm.num_public_inputs = b.public_inputs.size();
    if (!prover_data) { cd.pi_row = pi_row; return; }
    // subgroup, k_is (cosets.rs:9-24), sigma polynomials (permutation_argument.rs)
    cd.subgroup.resize(degree);
    { u64 g = primitive_root_of_unity(degree_bits), x = 1; for (size_t i = 0; i < degree; i++) { cd.subgroup[i] = x; x = mul(x, g); } }
    cd.num_targets = degree * cfg.num_wires + b.virtual_target_index;
    std::vector<size_t>& parent = cd.representative;
    parent.resize(cd.num_targets);
    std::iota(parent.begin(), parent.end(), size_t(0));
    auto tindex = [&](const Target& t) { return t.is_wire ? t.row * cfg.num_wires + t.col : degree * cfg.num_wires + t.index; };
    auto find = [&](size_t x) {
        size_t r = x; while (parent[r] != r) r = parent[r];
        while (parent[x] != x) { size_t o = parent[x]; parent[x] = r; x = o; }
        return r;
    };
    for (auto& cc : b.copy_constraints) { size_t x = find(tindex(cc.first)), y = find(tindex(cc.second)); if (x != y) parent[y] = x; }
    for (size_t i = 0; i < parent.size(); i++) find(i);
    // neighbours: within a class, routed wires in (row, column) order, cyclic
    std::vector<size_t> sigma(cfg.num_routed_wires * degree);
    {
        std::vector<size_t> first(cd.num_targets, SIZE_MAX), last(cd.num_targets, SIZE_MAX);
        for (size_t row = 0; row < degree; row++)
            for (size_t col = 0; col < cfg.num_routed_wires; col++) {
                size_t rep = parent[row * cfg.num_wires + col];
                size_t me = col * degree + row;
                if (first[rep] == SIZE_MAX) first[rep] = me; else sigma[last[rep]] = me;
                last[rep] = me;
            }
        for (size_t rep = 0; rep < cd.num_targets; rep++) if (first[rep] != SIZE_MAX) sigma[last[rep]] = first[rep];
    }
    std::vector<std::vector<u64>> sigma_vecs(cfg.num_routed_wires, std::vector<u64>(degree));
    for (size_t c = 0; c < cfg.num_routed_wires; c++)
        for (size_t r = 0; r < degree; r++) { size_t x = sigma[c * degree + r]; sigma_vecs[c][r] = mul(cm.k_is[x / degree], cd.subgroup[x % degree]); }

    cd.constants_sigmas = constant_vecs;
    for (auto& v : sigma_vecs) cd.constants_sigmas.push_back(v);
    cd.constants_sigmas_commitment = batch_from_values(cd.constants_sigmas, cfg.rate_bits, cfg.cap_height, threads);   // :1020-1028

    // circuit digest (:1089-1100): hash_no_pad(cap || hash_pad(domain_separator = []) || [degree_bits])
    {
        std::vector<u64> parts;
        for (auto& d : cd.constants_sigmas_commitment.tree.cap()) for (int k = 0; k < 4; k++) parts.push_back(canon(d.e[k]));
        std::vector<u64> padded = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};      // config.rs:41-51 pad10*1 of the empty message
        Digest ds = hash_no_pad(padded.data(), padded.size());
        for (int k = 0; k < 4; k++) parts.push_back(canon(ds.e[k]));
        parts.push_back(degree_bits);
        cd.circuit_digest = hash_no_pad(parts.data(), parts.size());
    }
    cd.row_gate.resize(degree);
    for (size_t j = 0; j < degree; j++) cd.row_gate[j] = b.gate_instances[j].type;
    cd.public_inputs = b.public_inputs;
    cd.arith_ops = b.arith_ops;
    cd.poseidon_rows = b.poseidon_rows;
    for (auto& so : b.split_ops) cd.split_ops.push_back({so.integer, so.rows});
    cd.exp_rows = b.exp_rows;
    for (auto& op : b.ra_ops) cd.ra_ops.push_back({op.row, op.copy, op.bits});
    for (auto& lr : b.lookup_rows) cd.lookup_rows.push_back({lr.last_lu_gate, lr.last_lut_gate, lr.first_lut_gate});
    cd.lut_to_lookups = b.lut_to_lookups;
    for (auto& ls : b.lookup_slots) cd.lookup_slots.push_back({ls.row, ls.slot, ls.lut});
    cd.pi_row = pi_row;
}

// Builds the m x m matmul circuit exactly as the demo does (plonky2/src/bin/matrix_mul.rs:25-67), then runs build().
static inline CircuitData build_matmul_circuit(size_t m, unsigned threads = 1, bool prover_data = true) {
    CircuitBuilder b;
    b.record_copies = prover_data;
    CircuitData cd;
    cd.m = m;
    std::vector<std::vector<Target>> A(m), B(m), C(m);
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < m; j++) { A[i].push_back(b.add_virtual_target()); B[i].push_back(b.add_virtual_target()); }
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < m; j++) {
            Target cur = b.constant(0);
            for (size_t k = 0; k < m; k++) { Target p = b.mul_t(A[i][k], B[k][j]); cur = b.add_t(cur, p); }
            C[i].push_back(cur);
        }
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < m; j++) { b.public_inputs.push_back(A[i][j]); b.public_inputs.push_back(B[i][j]); b.public_inputs.push_back(C[i][j]); }
    for (size_t i = 0; i < m; i++) for (size_t j = 0; j < m; j++) { cd.a_targets.push_back(A[i][j]); cd.b_targets.push_back(B[i][j]); }
    finish_build(b, cd, threads, prover_data);
    return cd;
}

// More circuits over the same gate set (tests only): shapes the matmul family never produces, and the reference's own example
// programs that use only these five gates (kinds 3-6: plonky2/examples/{fibonacci,factorial,easy_polynomial,square_root}.rs):
//   kind 1 "hash only": `param` inputs, all registered as public inputs -> gates {Noop, Constant, PublicInput, Poseidon}
//                       (no ArithmeticGate; selector groups {0,1,2} | {3})
//   kind 2 "chain":     x0, x1 inputs, `param` steps x_{i+2} = x_{i+1} x_i + x_i, NO public inputs -> gates {Noop, Constant,
//                       PublicInput, Arithmetic}: one selector group (num_selectors = 1, no UNUSED factor in the filters)
static inline CircuitData build_test_circuit(int kind, size_t param, unsigned threads = 1, bool prover_data = true) {
    CircuitBuilder b;
    b.record_copies = prover_data;
    CircuitData cd;
    cd.m = 0;
    if (kind == 1) {
        for (size_t i = 0; i < param; i++) { Target t = b.add_virtual_target(); cd.a_targets.push_back(t); b.public_inputs.push_back(t); }
    } else if (kind == 3) {
        // plonky2/examples/fibonacci.rs:21-36: `param` (99) additions from (initial_a, initial_b); public inputs a, b, result
        Target ia = b.add_virtual_target(), ib = b.add_virtual_target();
        Target prev = ia, cur = ib;
        for (size_t i = 0; i < param; i++) { Target t = b.add_t(prev, cur); prev = cur; cur = t; }
        b.public_inputs = {ia, ib, cur};
        cd.a_targets = {ia, ib};
    } else if (kind == 4) {
        // plonky2/examples/factorial.rs:22-33: cur = initial * 2 * 3 * ... * param (100); public inputs initial, result
        Target initial = b.add_virtual_target(), cur = initial;
        for (size_t i = 2; i <= param; i++) { Target it = b.constant((u64)i); cur = b.mul_t(cur, it); }
        b.public_inputs = {initial, cur};
        cd.a_targets = {initial};
    } else if (kind == 5) {
        // plonky2/examples/easy_polynomial.rs:19-28: x^2 - 4 x + 7; public inputs x, result
        Target x = b.add_virtual_target();
        Target a = b.mul_t(x, x);
        Target bb = b.mul_const(4, x);
        Target c = b.mul_const(GL_P - 1, bb);
        Target d = b.add_t(a, c);
        Target e = b.add_const(d, 7);
        b.public_inputs = {x, e};
        cd.a_targets = {x};
    } else if (kind == 8 || kind == 9) {
        // plonky2/src/lookup_test.rs:45-103 (test_one_lookup) / :276-362 (test_many_lookups): `param` inputs, each looked up in ONE table;
        // public inputs = the inputs, then the outputs.  kind 8: a 256-entry table on inputs 0..255 (the tests use the Tip5 S-box table;
        // any table exercises the same code, this one is out = (3 i^2 + 5 i + 7) mod 256); kind 9: a 10-entry table whose inputs are
        // not their indices (the generator's search path, lookup.rs:160-168)
        std::vector<std::pair<uint16_t, uint16_t>> table;
        if (kind == 8) for (unsigned i = 0; i < 256; i++) table.push_back({(uint16_t)i, (uint16_t)((3 * i * i + 5 * i + 7) % 256)});
        else for (unsigned i = 0; i < 10; i++) table.push_back({(uint16_t)(1000 + 37 * i), (uint16_t)(17 * i * i + 3)});
        const size_t ti = b.add_lookup_table_from_pairs(table);
        std::vector<Target> ins, outs;
        for (size_t i = 0; i < param; i++) { Target t = b.add_virtual_target(); ins.push_back(t); outs.push_back(b.add_lookup_from_index(t, ti)); }
        b.public_inputs = ins;
        b.public_inputs.insert(b.public_inputs.end(), outs.begin(), outs.end());
        cd.a_targets = ins;
    } else if (kind == 10 || kind == 11 || kind == 12) {
        // SEVERAL tables.  kind 10: plonky2/src/lookup_test.rs:107-187 (test_two_luts): two 256-entry tables, `param` lookups in each, and
        // the sum of the two first outputs through an ArithmeticGate.  kind 11: :191-271 (test_different_inputs): an 8-entry table on
        // inputs 2..9 next to a 256-entry one.  kind 12: :366-440 (test_same_luts): the same table added twice is stored once.
        // Inputs: `param` values for the first table, then `param` for the second.  Public inputs: inputs, outputs, (kind 10) the sum.
        std::vector<std::pair<uint16_t, uint16_t>> t0, t1;
        for (unsigned i = 0; i < 256; i++) t1.push_back({(uint16_t)i, (uint16_t)((3 * i * i + 5 * i + 7) % 256)});
        if (kind == 10) for (unsigned i = 0; i < 256; i++) t0.push_back({(uint16_t)i, (uint16_t)((7 * i + 1) % 256)});
        else if (kind == 11) for (unsigned i = 2; i < 10; i++) t0.push_back({(uint16_t)i, (uint16_t)(i * i + 1)});
        else t0 = t1;
        const size_t i0 = b.add_lookup_table_from_pairs(t0), i1 = b.add_lookup_table_from_pairs(t1);
        std::vector<Target> ins, outs;
        for (size_t i = 0; i < 2 * param; i++) { Target t = b.add_virtual_target(); ins.push_back(t); outs.push_back(b.add_lookup_from_index(t, i < param ? i0 : i1)); }
        b.public_inputs = ins;
        b.public_inputs.insert(b.public_inputs.end(), outs.begin(), outs.end());
        if (kind == 10) b.public_inputs.push_back(b.add_t(outs[0], outs[param]));
        cd.a_targets = ins;
    } else if (kind == 16) {
        // gadgets/random_access.rs:14-47 (its test_random_access_given_len): a list of 2^param targets, `reps` accesses of it at different
        // indices (reps = 3: several copies of one RandomAccessGate(param)), each result a public input.  Inputs: the list, then the indices.
        const size_t len = size_t(1) << param, reps = 3;
        std::vector<Target> v, idx, out;
        for (size_t i = 0; i < len; i++) v.push_back(b.add_virtual_target());
        for (size_t r = 0; r < reps; r++) idx.push_back(b.add_virtual_target());
        for (size_t r = 0; r < reps; r++) out.push_back(b.random_access(idx[r], v));
        b.public_inputs = idx;
        b.public_inputs.insert(b.public_inputs.end(), out.begin(), out.end());
        cd.a_targets = v;
        cd.a_targets.insert(cd.a_targets.end(), idx.begin(), idx.end());
    } else if (kind == 17) {
        // hash/merkle_proofs.rs:93-150 verify_merkle_proof_to_cap with a cap of FOUR digests (cap_height 2), the circuit of its
        // test_recursive_merkle_proof: `param` index bits, param - 2 siblings, cap_index = le_sum of the top two bits, four random
        // accesses (RandomAccessGate(2)) into the cap.  Inputs: leaf (5), index, siblings, cap (16); public inputs: cap and index.
        assert(param >= 2);
        std::vector<Target> leaf, ins;
        for (int i = 0; i < 5; i++) leaf.push_back(b.add_virtual_target());
        Target index = b.add_virtual_target();
        std::vector<std::array<Target, 4>> sibs(param - 2), cap(4);
        for (auto& sb : sibs) for (auto& t : sb) t = b.add_virtual_target();
        for (auto& c : cap) for (auto& t : c) t = b.add_virtual_target();
        std::vector<Target> bits = b.split_le(index, param);
        Target z = b.zero();
        std::array<Target, 4> state = b.hash_or_noop(leaf);
        for (size_t l = 0; l < sibs.size(); l++) {
            std::array<Target, 12> in; in.fill(z);
            for (int i = 0; i < 4; i++) { in[i] = state[i]; in[4 + i] = sibs[l][i]; }
            auto out = b.permute_swapped(in, bits[l]);
            for (int i = 0; i < 4; i++) state[i] = out[i];
        }
        Target cap_index = b.le_sum(std::vector<Target>(bits.begin() + sibs.size(), bits.end()));
        for (int i = 0; i < 4; i++) {
            Target r = b.random_access(cap_index, {cap[0][i], cap[1][i], cap[2][i], cap[3][i]});
            b.connect(r, state[i]);
        }
        ins = leaf; ins.push_back(index);
        for (auto& sb : sibs) for (auto& t : sb) ins.push_back(t);
        for (auto& c : cap) for (auto& t : c) { ins.push_back(t); b.public_inputs.push_back(t); }
        b.public_inputs.push_back(index);
        cd.a_targets = ins;
    } else if (kind == 15) {
        // every supported gate type in one circuit (11 gate types: Noop, Constant, PublicInput, Arithmetic, Poseidon, BaseSum<2>, Exponentiation,
        // and a LookupGate + LookupTableGate for each of two tables): several selector groups.  Inputs: x (< 256), y (< 2^param, param <= 16).
        // out0 = table0[x], out1 = table1[x], s = out0 + out1, r = (s + 3)^y, y range-checked to `param` bits; public inputs x, y, s, r.
        std::vector<std::pair<uint16_t, uint16_t>> t0, t1;
        for (unsigned i = 0; i < 256; i++) { t0.push_back({(uint16_t)i, (uint16_t)((7 * i + 1) % 256)}); t1.push_back({(uint16_t)i, (uint16_t)((3 * i * i + 5 * i + 7) % 256)}); }
        const size_t i0 = b.add_lookup_table_from_pairs(t0), i1 = b.add_lookup_table_from_pairs(t1);
        Target x = b.add_virtual_target(), y = b.add_virtual_target();
        Target o0 = b.add_lookup_from_index(x, i0), o1 = b.add_lookup_from_index(x, i1);
        Target sum = b.add_t(o0, o1);
        Target r = b.exp(b.add_const(sum, 3), y, param);
        b.public_inputs = {x, y, sum, r};
        cd.a_targets = {x, y};
    } else if (kind == 14) {
        // hash/merkle_proofs.rs:78-150, as in its test_recursive_merkle_proof (:175-220) with a root instead of a cap: a 5-element leaf at a
        // `param`-bit index, `param` sibling digests, the root.  Inputs: leaf (5), index, siblings (4 x param), root (4); public inputs: the
        // root and the index.  The PoseidonGates run with swap = the index bits (the demo circuit only ever has swap = 0).
        std::vector<Target> leaf, ins;
        for (int i = 0; i < 5; i++) leaf.push_back(b.add_virtual_target());
        Target index = b.add_virtual_target();
        std::vector<std::array<Target, 4>> sibs(param);
        for (auto& sb : sibs) for (auto& t : sb) t = b.add_virtual_target();
        std::array<Target, 4> root; for (auto& t : root) t = b.add_virtual_target();
        b.verify_merkle_proof(leaf, b.split_le(index, param), root, sibs);
        ins = leaf; ins.push_back(index);
        for (auto& sb : sibs) for (auto& t : sb) ins.push_back(t);
        for (auto& t : root) ins.push_back(t);
        b.public_inputs = {root[0], root[1], root[2], root[3], index};
        cd.a_targets = ins;
    } else if (kind == 13) {
        // gadgets/arithmetic.rs:268-272 exp(base, exponent, num_bits = param): public inputs base, exponent, base^exponent; the exponent is
        // split by a BaseSumGate<2>, the power taken by an ExponentiationGate (the first gate of degree 4 in these circuits)
        Target base = b.add_virtual_target(), e = b.add_virtual_target();
        Target r = b.exp(base, e, param);
        b.public_inputs = {base, e, r};
        cd.a_targets = {base, e};
    } else if (kind == 7) {
        // plonky2/examples/range_check.rs:20-24: the value is a public input and is range-checked to `param` (6) bits
        Target value = b.add_virtual_target();
        b.public_inputs = {value};
        b.range_check(value, param);
        cd.a_targets = {value};
    } else if (kind == 6) {
        // plonky2/examples/square_root.rs:104-107: x_squared = square(x) is the public input; the example's SquareRootGenerator finds
        // x from x_squared outside the circuit -- at the witness-matrix boundary x is simply given
        Target x = b.add_virtual_target();
        Target x2 = b.mul_t(x, x);                          // square(x) = mul(x, x) (arithmetic.rs:199-201)
        b.public_inputs = {x2};
        cd.a_targets = {x};
    } else {
        Target x0 = b.add_virtual_target(), x1 = b.add_virtual_target();
        cd.a_targets = {x0, x1};
        for (size_t i = 0; i < param; i++) { Target x2 = b.arithmetic(1, 1, x1, x0, x0); x0 = x1; x1 = x2; }
    }
    finish_build(b, cd, threads, prover_data);
    return cd;
}

// ---- witness -------------------------------------------------------------------------------------------
static inline u64 splitmix64_next(u64& x) {
    x += 0x9E3779B97F4A7C15ULL;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// PoseidonGenerator (gates/poseidon.rs:430-497): fills delta, s-box-input and output wires of one row
static inline void poseidon_gate_witness(const u64* in12, u64 swap, u64* row135) {
    typedef PoseidonWires PW;
    u64 s[12];
    for (int i = 0; i < 12; i++) { s[i] = in12[i]; row135[PW::INPUT + i] = in12[i]; }
    row135[PW::SWAP] = swap;
    for (int i = 0; i < 4; i++) row135[PW::DELTA + i] = mul(swap, sub(s[i + 4], s[i]));
    if (canon(swap) == 1) for (int i = 0; i < 4; i++) std::swap(s[i], s[i + 4]);
    int round = 0;
    for (int r = 0; r < 4; r++) {
        pk_constant_layer<u64>(s, round);
        if (r != 0) for (int i = 0; i < 12; i++) row135[PW::full_sbox_0(r, i)] = s[i];
        for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        pk_mds_layer<u64>(s);
        round++;
    }
    pk_partial_init<u64>(s);
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {
        row135[PW::partial_sbox(r)] = s[0];
        s[0] = add(sbox7(s[0]), POSEIDON_PARTIAL_RC[r]);
        pk_partial_fast<u64>(s, r);
    }
    round += POSEIDON_PARTIAL_ROUNDS;
    for (int r = 0; r < 4; r++) {
        pk_constant_layer<u64>(s, round);
        for (int i = 0; i < 12; i++) row135[PW::full_sbox_1(r, i)] = s[i];
        for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        pk_mds_layer<u64>(s);
        round++;
    }
    for (int i = 0; i < 12; i++) row135[PW::OUTPUT + i] = s[i];
}

struct Witness {
    std::vector<std::vector<u64>> wire_values;    // [135][n], canonical (iop/witness.rs:256-258)
    std::vector<u64> public_inputs;
};

// a, b: row-major m x m inputs.  `filler_seed` drives the 131 values the reference draws from OsRng for the
// unused PublicInputGate wires (circuit_builder.rs:904-910): the witness MATRIX is the parity boundary.
static inline Witness generate_witness(const CircuitData& cd, const std::vector<u64>& a, const std::vector<u64>& b, u64 filler_seed) {
    const size_t nw = cd.common.config.num_wires, degree = cd.common.degree();
    std::vector<u64> val(cd.num_targets, 0);
    std::vector<char> set(cd.num_targets, 0);
    auto rep = [&](const Target& t) { return cd.representative[cd.target_index(t)]; };
    auto put = [&](const Target& t, u64 v) { size_t r = rep(t); assert(!set[r] || canon(val[r]) == canon(v)); val[r] = canon(v); set[r] = 1; };
    auto get = [&](const Target& t) { size_t r = rep(t); assert(set[r]); return val[r]; };
    assert(a.size() == cd.a_targets.size() && b.size() == cd.b_targets.size());
    for (size_t i = 0; i < a.size(); i++) put(cd.a_targets[i], a[i]);
    for (size_t i = 0; i < b.size(); i++) put(cd.b_targets[i], b[i]);
    for (auto& cw : cd.constant_wires) put(Target::wire(cw.row, cw.wire), cw.value);
    // The reference runs every generator once its dependencies are known (iop/generator.rs:19-98): ONE fixed point over all the kinds
    // of generator present here (a lookup output may feed arithmetic, an exponentiation, a hash ...).  The order cannot change a value.
    auto is_set = [&](const Target& t) { return set[rep(t)] != 0; };
    // LookupTableGenerator (lookup_table.rs:175-203): no dependencies; table entries fill the LUT rows upside down, padded with zeros
    for (size_t li = 0; li < cd.lookup_rows.size(); li++) {
        auto& lr = cd.lookup_rows[li]; auto& lut = cd.common.luts[li];
        for (size_t row = lr.last_lut_gate; row <= lr.first_lut_gate; row++)
            for (size_t sl = 0; sl < LOOKUP_TABLE_SLOTS; sl++) {
                const size_t entry = (lr.first_lut_gate - row) * LOOKUP_TABLE_SLOTS + sl;
                put(Target::wire(row, 3 * sl), entry < lut.size() ? lut[entry].first : 0);
                put(Target::wire(row, 3 * sl + 1), entry < lut.size() ? lut[entry].second : 0);
            }
    }
    std::vector<char> ra_done(cd.ra_ops.size(), 0);
    std::vector<char> lk_done(cd.lookup_slots.size(), 0);      // LookupGenerator (lookup.rs:151-175): one per slot, needs its looking input
    std::vector<char> arith_done(cd.arith_ops.size(), 0), split_done(cd.split_ops.size(), 0), pos_done(cd.poseidon_rows.size(), 0), exp_done(cd.exp_rows.size(), 0);
    std::vector<u64> rowbuf(135);
    for (bool progress = true; progress;) {
        progress = false;
        for (size_t k = 0; k < cd.arith_ops.size(); k++) {   // ArithmeticBaseGenerator (arithmetic_base.rs:203-218)
            auto& op = cd.arith_ops[k];
            Target t0 = Target::wire(op.row, 4 * op.slot), t1 = Target::wire(op.row, 4 * op.slot + 1), t2 = Target::wire(op.row, 4 * op.slot + 2);
            if (arith_done[k] || !is_set(t0) || !is_set(t1) || !is_set(t2)) continue;
            put(Target::wire(op.row, 4 * op.slot + 3), add(mul(mul(get(t0), get(t1)), op.c0), mul(get(t2), op.c1)));
            arith_done[k] = 1; progress = true;
        }
        for (size_t k = 0; k < cd.split_ops.size(); k++) {   // WireSplitGenerator (split_join.rs:117-142), BaseSplitGenerator (base_sum.rs:183-207)
            auto& so = cd.split_ops[k];
            if (split_done[k] || !is_set(so.integer)) continue;
            u64 v = canon(get(so.integer));
            for (size_t row : so.rows) {
                u64 t = v & ((u64(1) << BASE_SUM_LIMBS) - 1);
                v >>= BASE_SUM_LIMBS;
                put(Target::wire(row, 0), t);
                for (size_t i = 0; i < BASE_SUM_LIMBS; i++) { put(Target::wire(row, 1 + i), t & 1); t >>= 1; }
            }
            assert(v == 0 && "Integer too large to fit in the BaseSumGates");
            split_done[k] = 1; progress = true;
        }
        for (size_t k = 0; k < cd.lookup_slots.size(); k++) {
            auto& ls = cd.lookup_slots[k];
            Target tin = Target::wire(ls.row, 2 * ls.slot);
            if (lk_done[k] || !is_set(tin)) continue;
            const u64 v = canon(get(tin));
            auto& lut = cd.common.luts[ls.lut];
            size_t idx = 0;
            if (v < lut.size() && lut[v].first == v) idx = v;
            else { while (idx < lut.size() && lut[idx].first != v) idx++; assert(idx < lut.size() && "Incorrect input value provided"); }
            put(Target::wire(ls.row, 2 * ls.slot + 1), lut[idx].second);
            lk_done[k] = 1; progress = true;
        }
        for (size_t k = 0; k < cd.ra_ops.size(); k++) {       // RandomAccessGenerator (random_access.rs:335-390)
            auto& op = cd.ra_ops[k];
            const RandomAccess ra(op.bits);
            if (ra_done[k]) continue;
            bool ready = is_set(Target::wire(op.row, ra.wire_access_index(op.copy)));
            for (size_t i = 0; i < ra.vec_size && ready; i++) ready = is_set(Target::wire(op.row, ra.wire_list_item(i, op.copy)));
            if (!ready) continue;
            const u64 idx = canon(get(Target::wire(op.row, ra.wire_access_index(op.copy))));
            assert(idx < ra.vec_size && "Access index is larger than the vector size");
            put(Target::wire(op.row, ra.wire_claimed_element(op.copy)), get(Target::wire(op.row, ra.wire_list_item(idx, op.copy))));
            for (size_t i = 0; i < ra.bits; i++) put(Target::wire(op.row, ra.wire_bit(i, op.copy)), (idx >> i) & 1);
            ra_done[k] = 1; progress = true;
        }
        for (size_t k = 0; k < cd.exp_rows.size(); k++) {     // ExponentiationGenerator (exponentiation.rs:233-280)
            const size_t row = cd.exp_rows[k], n = EXP_POWER_BITS;
            if (exp_done[k]) continue;
            bool ready = is_set(Target::wire(row, 0));
            for (size_t i = 0; i < n && ready; i++) ready = is_set(Target::wire(row, 1 + i));
            if (!ready) continue;
            const u64 base = get(Target::wire(row, 0));
            u64 cur = 1;
            for (size_t i = 0; i < n; i++) {
                if (canon(get(Target::wire(row, 1 + (n - 1 - i)))) == 1) cur = mul(cur, base);
                put(Target::wire(row, 2 + n + i), cur);
                if (i + 1 == n) put(Target::wire(row, 1 + n), cur);
                cur = mul(cur, cur);
            }
            exp_done[k] = 1; progress = true;
        }
        for (size_t k = 0; k < cd.poseidon_rows.size(); k++) {
            const size_t row = cd.poseidon_rows[k];
            if (pos_done[k]) continue;
            bool ready = is_set(Target::wire(row, PoseidonWires::SWAP));
            for (int i = 0; i < 12 && ready; i++) ready = is_set(Target::wire(row, PoseidonWires::INPUT + i));
            if (!ready) continue;
            u64 in[12];
            for (int i = 0; i < 12; i++) in[i] = get(Target::wire(row, PoseidonWires::INPUT + i));
            u64 swap = get(Target::wire(row, PoseidonWires::SWAP));
            poseidon_gate_witness(in, swap, rowbuf.data());
            for (int c = PoseidonWires::DELTA; c < PoseidonWires::END; c++) put(Target::wire(row, c), rowbuf[c]);
            for (int i = 0; i < 12; i++) put(Target::wire(row, PoseidonWires::OUTPUT + i), rowbuf[PoseidonWires::OUTPUT + i]);
            pos_done[k] = 1; progress = true;
        }
    }
    for (char d : lk_done) assert(d && "a lookup never became ready");
    // set_lookup_wires (prover.rs:34-99): multiplicities of the table entries, the last LookupGate padded with the first entry
    for (size_t li = 0; li < cd.lookup_rows.size(); li++) {
        auto& lr = cd.lookup_rows[li]; auto& lut = cd.common.luts[li];
        std::vector<u64> mult(lut.size(), 0);
        for (auto& lk : cd.lut_to_lookups[li]) {
            const u64 v = canon(get(lk.first));
            size_t idx = 0;
            while (lut[idx].first != v) idx++;
            mult[idx]++;
        }
        const size_t remaining = (LOOKUP_SLOTS - (cd.lut_to_lookups[li].size() % LOOKUP_SLOTS)) % LOOKUP_SLOTS;
        for (size_t sl = LOOKUP_SLOTS - remaining; sl < LOOKUP_SLOTS; sl++) {
            put(Target::wire(lr.last_lut_gate - 1, 2 * sl), lut[0].first);
            put(Target::wire(lr.last_lut_gate - 1, 2 * sl + 1), lut[0].second);
            mult[0]++;
        }
        for (size_t e = 0; e < lut.size(); e++)
            put(Target::wire(lr.first_lut_gate - e / LOOKUP_TABLE_SLOTS, 3 * (e % LOOKUP_TABLE_SLOTS) + 2), mult[e]);
    }
    for (char d : arith_done) assert(d && "an arithmetic operation never became ready");
    for (char d : pos_done) assert(d && "a PoseidonGate never became ready");
    { u64 st = filler_seed; for (size_t c = 4; c < nw; c++) put(Target::wire(cd.pi_row, c), splitmix64_next(st) % GL_P); }
    Witness w;
    w.wire_values.assign(nw, std::vector<u64>(degree, 0));
    for (size_t i = 0; i < degree; i++)
        for (size_t j = 0; j < nw; j++) { size_t r = cd.representative[i * nw + j]; if (set[r]) w.wire_values[j][i] = val[r]; }
    for (auto& t : cd.public_inputs) w.public_inputs.push_back(get(t));
    return w;
}

}  // namespace orc
