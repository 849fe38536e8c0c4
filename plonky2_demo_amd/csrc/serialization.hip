// CommonCircuitData / VerifierOnlyCircuitData <-> bytes, the wire format of the reference's circuit serialisation
// (plonky2/src/util/serialization/mod.rs:1596-1790 write_fri_* / write_circuit_config / write_selectors_info /
// write_common_circuit_data, :1889-1906 write_verifier_only_circuit_data, :1908-1919 write_verifier_circuit_data; gate tags as in
// DefaultGateSerializer, util/serialization/gate_serialization.rs:87-108; per-gate payloads gates/arithmetic_base.rs:63-65,
// gates/constant.rs:49-51, none for Noop / Poseidon / PublicInput).  Host code.
//
// The reference ships no serialised circuit, so these bytes cannot be compared with a Rust-written file here ("parity
// unpinned"); what is checked is a round trip, a second independent writer in the test oracle, and that
// gl_verify_bytes accepts proofs through data that went through the byte form.
//
// Only circuits this library can prove / verify are representable: the five gates of the demo + BaseSumGate<2>, standard_recursion_config's
// shape (no lookups, no zero-knowledge); anything else is GL_ERR_UNSUPPORTED when reading.
#include "context.hpp"
#include "host_circuit.hpp"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {
// position of each gate type in DefaultGateSerializer's list (gate_serialization.rs:89-107)
const uint32_t TAG_ARITHMETIC = 0, TAG_BASE_SUM_2 = 2, TAG_CONSTANT = 3, TAG_EXPONENTIATION = 5, TAG_LOOKUP = 6, TAG_LOOKUP_TABLE = 7, TAG_NOOP = 9, TAG_POSEIDON = 11, TAG_PUBLIC_INPUT = 12, TAG_RANDOM_ACCESS = 13;
const uint64_t LOOKUP_SLOTS = 40, LOOKUP_TABLE_SLOTS = 26;     // gates/lookup.rs:41-44, gates/lookup_table.rs:47-50
const uint64_t BASE_SUM_LIMBS = 63;     // BaseSumGate::<2>::new_from_config under standard_recursion_config (gates/base_sum.rs:31-35)
// standard_recursion_config (plonk/circuit_data.rs:72-90)
const uint64_t STD_SECURITY_BITS = 100, STD_CONFIG_NUM_CONSTANTS = 2, STD_FINAL_POLY_BITS = 5, STD_ARITY_BITS = 4;

struct Writer {
    std::vector<uint8_t> b;
    void u8(uint8_t x) { b.push_back(x); }
    void u32(uint32_t x) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void u64(uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }       // write_usize (mod.rs:1220-1222)
    void field(gl_t x) { u64(gl_canon(x)); }                                                          // write_field (mod.rs:1237-1242)
    void u16(uint16_t x) { b.push_back((uint8_t)x); b.push_back((uint8_t)(x >> 8)); }
    void lut(const gl_circuit_desc& d, unsigned t) {                                                 // write_lut (mod.rs:2077-2085)
        const uint16_t* e = d.lut + 2 * (size_t)glhost::lut_offset(d, t);
        u64(d.lut_len[t]); for (uint32_t i = 0; i < 2 * d.lut_len[t]; i++) u16(e[i]);
    }
};
struct Reader {
    const uint8_t* p; size_t n, pos = 0; bool ok = true;
    Reader(const uint8_t* b, size_t len) : p(b), n(len) {}
    uint64_t take(int k) { if (!ok || n - pos < (size_t)k) { ok = false; return 0; } uint64_t v = 0; for (int i = 0; i < k; i++) v |= (uint64_t)p[pos + i] << (8 * i); pos += k; return v; }
    uint8_t u8() { return (uint8_t)take(1); }
    uint32_t u32() { return (uint32_t)take(4); }
    uint16_t u16() { return (uint16_t)take(2); }
    uint64_t u64() { return take(8); }
    // a `usize` that this library keeps in 32 bits: anything larger is a file this library cannot represent, never a narrowing
    // (2^32 + 135 must not read as 135)
    uint32_t usize32() { const uint64_t v = take(8); if (v > 0xFFFFFFFFull) { ok = false; wide = true; return 0; } return (uint32_t)v; }
    bool wide = false;
};

uint32_t gate_tag(uint8_t type) {
    switch (type) { case 0: return TAG_NOOP; case 1: return TAG_CONSTANT; case 2: return TAG_PUBLIC_INPUT; case 3: return TAG_ARITHMETIC; case 5: return TAG_BASE_SUM_2; case 6: return TAG_LOOKUP; case 7: return TAG_LOOKUP_TABLE; case 8: return TAG_EXPONENTIATION; case 9: return TAG_RANDOM_ACCESS; default: return TAG_POSEIDON; }
}
uint64_t gate_constraints(uint8_t type, uint8_t param, const gl_circuit_desc& d) {
    if (type == 9) return glhost::RandomAccessLayout(param).num_constraints();
    switch (type) { case 0: case 6: case 7: return 0; case 1: return d.num_constants - d.num_selectors - d.num_lookup_selectors; case 2: return 4; case 3: return d.num_routed_wires / 4; case 5: return 1 + BASE_SUM_LIMBS /* gates/base_sum.rs:144-146 */; case 8: return glhost::EXP_POWER_BITS + 1 /* gates/exponentiation.rs:190-192 */; default: return 123; }   // gates/poseidon.rs:403-409
}
void write_fri_config(Writer& w, const gl_circuit_desc& d) {          // mod.rs:1628-1644
    w.u64(d.rate_bits); w.u64(d.cap_height); w.u64(d.num_query_rounds); w.u32(d.proof_of_work_bits);
    w.u8(1); w.u64(STD_ARITY_BITS); w.u64(STD_FINAL_POLY_BITS);       // FriReductionStrategy::ConstantArityBits(4, 5) (mod.rs:1609-1615)
}
int read_fri_config(Reader& r, gl_circuit_desc& d) {
    d.rate_bits = r.usize32(); d.cap_height = r.usize32(); d.num_query_rounds = r.usize32(); d.proof_of_work_bits = r.u32();
    const uint8_t strat = r.u8();
    // the reduction strategy is not part of gl_circuit_desc (the rounds themselves are): only the one this library writes back
    // is accepted, so that bytes -> desc -> bytes cannot silently change a field
    if (strat == 1) {
        const uint64_t arity = r.u64(), final_bits = r.u64();
        GL_REQUIRE(!r.ok || (arity == STD_ARITY_BITS && final_bits == STD_FINAL_POLY_BITS), GL_ERR_UNSUPPORTED, "FRI reduction strategy other than ConstantArityBits(4, 5)");
    } else if (strat == 0 || strat == 2) return gl_fail(GL_ERR_UNSUPPORTED, "FRI reduction strategy other than ConstantArityBits(4, 5)", __FILE__, __LINE__);
    else return gl_fail(GL_ERR_ARG, "unknown FRI reduction strategy", __FILE__, __LINE__);
    return GL_OK;
}
}   // namespace

// CommonCircuitData -> bytes.  *num_bytes receives the size; h_out may be null to query it.
extern "C" int gl_common_data_to_bytes(const gl_circuit_desc* desc, uint8_t* h_out, size_t cap, size_t* num_bytes) {
    GL_REQUIRE(desc && num_bytes, GL_ERR_ARG, "gl_common_data_to_bytes: null argument");
    const gl_circuit_desc& d = *desc;
    GL_REQUIRE(d.num_gates >= 1 && d.num_gates <= GL_MAX_GATES && d.num_fri_rounds <= 8 && d.num_selectors >= 1 && d.num_selectors <= 4, GL_ERR_ARG, "bad circuit description");
    // everything the writer indexes or divides by (k_is holds 80 entries)
    GL_REQUIRE(d.num_routed_wires >= 4 && d.num_routed_wires <= 80 && d.quotient_degree_factor >= 1 && d.num_constants >= d.num_selectors, GL_ERR_ARG,
               "bad circuit description: routed wires 4..80, quotient degree factor >= 1, constants >= selectors");
    for (uint32_t g = 0; g < d.num_gates; g++) GL_REQUIRE(d.gate_types[g] <= glhost::G_LAST, GL_ERR_UNSUPPORTED, "gate type not in {Noop, Constant, PublicInput, Arithmetic, Poseidon, BaseSum<2>, Lookup, LookupTable, Exponentiation, RandomAccess}");
    { const char* why = glhost::lookup_shape_error(d); GL_REQUIRE(!why, GL_ERR_ARG, why); }
    GL_REQUIRE(d.num_constants >= d.num_selectors + d.num_lookup_selectors, GL_ERR_ARG, "bad lookup description");
    Writer w;
    // CircuitConfig (mod.rs:1662-1686)
    w.u64(d.num_wires); w.u64(d.num_routed_wires); w.u64(STD_CONFIG_NUM_CONSTANTS); w.u64(STD_SECURITY_BITS); w.u64(d.num_challenges);
    w.u64(d.quotient_degree_factor); w.u8(1 /* use_base_arithmetic_gate */); w.u8(0 /* zero_knowledge */);
    write_fri_config(w, d);
    // FriParams (mod.rs:1646-1660)
    write_fri_config(w, d);
    w.u64(d.num_fri_rounds); for (uint32_t i = 0; i < d.num_fri_rounds; i++) w.u64(d.fri_arity_bits[i]);
    w.u64(d.degree_bits); w.u8(0 /* hiding */);
    // gates (mod.rs:1759-1762)
    w.u64(d.num_gates);
    uint64_t max_constraints = 0;
    for (uint32_t g = 0; g < d.num_gates; g++) {
        w.u32(gate_tag(d.gate_types[g]));
        if (d.gate_types[g] == 3) w.u64(d.num_routed_wires / 4);                   // ArithmeticGate { num_ops }
        if (d.gate_types[g] == 1) w.u64(d.num_constants - d.num_selectors - d.num_lookup_selectors);      // ConstantGate { num_consts }
        if (d.gate_types[g] == 6) { w.u64(LOOKUP_SLOTS); w.lut(d, d.gate_params[g]); }                // LookupGate { num_slots, lut } (gates/lookup.rs:59-62)
        if (d.gate_types[g] == 7) { w.u64(LOOKUP_TABLE_SLOTS); w.lut(d, d.gate_params[g]); w.u64(d.last_lut_row[d.gate_params[g]]); }      // LookupTableGate (gates/lookup_table.rs:70-74)
        if (d.gate_types[g] == 5) w.u64(BASE_SUM_LIMBS);                           // BaseSumGate<2> { num_limbs } (gates/base_sum.rs:53-55)
        if (d.gate_types[g] == 9) { const glhost::RandomAccessLayout ra(d.gate_params[g]); w.u64(ra.bits); w.u64(ra.num_copies); w.u64(ra.num_extra_constants); }      // random_access.rs:123-128
        if (d.gate_types[g] == 8) w.u64(glhost::EXP_POWER_BITS);                   // ExponentiationGate { num_power_bits } (gates/exponentiation.rs:79-81)
        const uint64_t c = gate_constraints(d.gate_types[g], d.gate_params[g], d);
        if (c > max_constraints) max_constraints = c;
    }
    // SelectorsInfo (mod.rs:1700-1713): selector_indices, then the distinct groups in order
    w.u64(d.num_gates); for (uint32_t g = 0; g < d.num_gates; g++) w.u64(d.gate_selector_index[g]);
    std::vector<std::pair<uint32_t, uint32_t>> groups;
    for (uint32_t g = 0; g < d.num_gates; g++) {
        const std::pair<uint32_t, uint32_t> gr(d.gate_group_start[g], d.gate_group_end[g]);
        if (groups.empty() || groups.back() != gr) groups.push_back(gr);
    }
    w.u64(groups.size()); for (auto& gr : groups) { w.u64(gr.first); w.u64(gr.second); }
    w.u64(d.quotient_degree_factor); w.u64(max_constraints); w.u64(d.num_constants); w.u64(d.num_public_inputs);
    w.u64(d.num_routed_wires); for (uint32_t j = 0; j < d.num_routed_wires; j++) w.field(d.k_is[j]);
    w.u64((d.num_routed_wires + d.quotient_degree_factor - 1) / d.quotient_degree_factor - 1);      // num_partial_products (circuit_builder.rs, util/partial_products.rs:40-47)
    w.u64(d.num_lookup_polys); w.u64(d.num_lookup_selectors);                      // mod.rs:1776-1782
    w.u64(d.num_luts); for (unsigned t = 0; t < d.num_luts; t++) w.lut(d, t);      // luts
    *num_bytes = w.b.size();
    if (!h_out) return GL_OK;
    GL_REQUIRE(cap >= w.b.size(), GL_ERR_ARG, "gl_common_data_to_bytes: output too small");
    memcpy(h_out, w.b.data(), w.b.size());
    return GL_OK;
}

// bytes -> CommonCircuitData (read_common_circuit_data, mod.rs:739-800).  *consumed receives the bytes read.
extern "C" int gl_common_data_from_bytes(const uint8_t* h_bytes, size_t num_bytes, gl_circuit_desc* out, size_t* consumed) {
    GL_REQUIRE(h_bytes && out, GL_ERR_ARG, "gl_common_data_from_bytes: null argument");
    Reader r(h_bytes, num_bytes);
    gl_circuit_desc d;
    memset(&d, 0, sizeof d);
    d.num_wires = r.usize32(); d.num_routed_wires = r.usize32();
    const uint64_t cfg_consts = r.u64(), security_bits = r.u64();
    d.num_challenges = r.usize32();
    const uint64_t max_qdf = r.u64();
    const uint8_t base_arith = r.u8();                  // use_base_arithmetic_gate
    const uint8_t zk = r.u8();
    GL_TRY(read_fri_config(r, d));                      // config.fri_config
    gl_circuit_desc fp = d;
    GL_TRY(read_fri_config(r, fp));                     // fri_params.config: the same values
    const uint64_t nred = r.u64();
    GL_REQUIRE(r.ok && nred <= 8, GL_ERR_UNSUPPORTED, "more than 8 FRI reduction rounds");
    d.num_fri_rounds = (uint32_t)nred;
    for (uint64_t i = 0; i < nred; i++) d.fri_arity_bits[i] = r.usize32();
    d.degree_bits = r.usize32();
    const uint8_t hiding = r.u8();
    GL_REQUIRE(!r.wide, GL_ERR_UNSUPPORTED, "a size field of CommonCircuitData exceeds 32 bits");
    GL_REQUIRE(r.ok, GL_ERR_ARG, "truncated CommonCircuitData");
    GL_REQUIRE(!zk && !hiding, GL_ERR_UNSUPPORTED, "zero-knowledge circuits are not supported");
    // fields gl_circuit_desc does not carry are written back as standard_recursion_config's: anything else is refused, not dropped
    GL_REQUIRE(security_bits == STD_SECURITY_BITS && base_arith == 1, GL_ERR_UNSUPPORTED, "security_bits / use_base_arithmetic_gate differ from standard_recursion_config");
    GL_REQUIRE(fp.rate_bits == d.rate_bits && fp.cap_height == d.cap_height && fp.num_query_rounds == d.num_query_rounds && fp.proof_of_work_bits == d.proof_of_work_bits,
               GL_ERR_ARG, "fri_params.config differs from config.fri_config");
    const uint64_t ngates = r.u64();
    GL_REQUIRE(r.ok && ngates >= 1 && ngates <= GL_MAX_GATES, GL_ERR_UNSUPPORTED, "1..16 gate types");
    d.num_gates = (uint32_t)ngates;
    uint64_t arith_ops = 0, const_consts = 0;
    // a lookup gate carries its whole table (lookup.rs:59-62, lookup_table.rs:70-74); which entry of `luts` it is gets settled below
    std::vector<uint16_t> gate_table[GL_MAX_GATES];
    uint32_t gate_last_lut_row[GL_MAX_GATES] = {0};
    for (uint64_t g = 0; g < ngates; g++) {
        const uint32_t tag = r.u32();
        if (tag == TAG_NOOP) d.gate_types[g] = 0;
        else if (tag == TAG_CONSTANT) { d.gate_types[g] = 1; const_consts = r.u64(); }
        else if (tag == TAG_PUBLIC_INPUT) d.gate_types[g] = 2;
        else if (tag == TAG_ARITHMETIC) { d.gate_types[g] = 3; arith_ops = r.u64(); }
        else if (tag == TAG_POSEIDON) d.gate_types[g] = 4;
        else if (tag == TAG_BASE_SUM_2) {
            d.gate_types[g] = 5;
            const uint64_t limbs = r.u64();
            GL_REQUIRE(!r.ok || limbs == BASE_SUM_LIMBS, GL_ERR_UNSUPPORTED, "BaseSumGate<2> with a limb count other than new_from_config's 63");
        }
        else if (tag == TAG_RANDOM_ACCESS) {
            d.gate_types[g] = 9;
            const uint64_t bits = r.u64(), copies = r.u64(), extra = r.u64();
            GL_REQUIRE(!r.ok || (bits >= 1 && bits <= 6), GL_ERR_UNSUPPORTED, "RandomAccessGate: 1..6 index bits");
            if (r.ok) {
                const glhost::RandomAccessLayout ra((uint32_t)bits);
                GL_REQUIRE(copies == ra.num_copies && extra == ra.num_extra_constants, GL_ERR_UNSUPPORTED, "RandomAccessGate with a layout other than new_from_config's");
                d.gate_params[g] = (uint8_t)bits;
            }
        }
        else if (tag == TAG_EXPONENTIATION) {
            d.gate_types[g] = 8;
            const uint64_t bits = r.u64();
            GL_REQUIRE(!r.ok || bits == glhost::EXP_POWER_BITS, GL_ERR_UNSUPPORTED, "ExponentiationGate with a bit count other than new_from_config's 66");
        }
        else if (tag == TAG_LOOKUP || tag == TAG_LOOKUP_TABLE) {
            d.gate_types[g] = tag == TAG_LOOKUP ? 6 : 7;
            const uint64_t slots = r.u64(), len = r.u64();
            GL_REQUIRE(!r.ok || (slots == (tag == TAG_LOOKUP ? LOOKUP_SLOTS : LOOKUP_TABLE_SLOTS) && len >= 1 && len <= GL_MAX_LUT_ENTRIES), GL_ERR_UNSUPPORTED,
                       "lookup gate: slot count of standard_recursion_config and a table of at most 1024 entries");
            for (uint64_t k = 0; k < 2 * len && r.ok; k++) gate_table[g].push_back(r.u16());
            if (tag == TAG_LOOKUP_TABLE) gate_last_lut_row[g] = r.usize32();
        }
        else return gl_fail(GL_ERR_UNSUPPORTED, "gate outside {Noop, Constant, PublicInput, Arithmetic, Poseidon, BaseSum<2>, Lookup, LookupTable, Exponentiation, RandomAccess}", __FILE__, __LINE__);
    }
    const uint64_t nsel = r.u64();
    GL_REQUIRE(r.ok && nsel == ngates, GL_ERR_ARG, "selector_indices length differs from the number of gates");
    for (uint64_t g = 0; g < ngates; g++) d.gate_selector_index[g] = r.usize32();
    const uint64_t ngroups = r.u64();
    GL_REQUIRE(r.ok && ngroups >= 1 && ngroups <= 4, GL_ERR_UNSUPPORTED, "1..4 selector groups");
    d.num_selectors = (uint32_t)ngroups;
    for (uint64_t k = 0; k < ngroups; k++) {
        const uint32_t start = r.usize32(), end = r.usize32();
        GL_REQUIRE(r.ok && start < end && end <= ngates, GL_ERR_ARG, "bad selector group");
        for (uint32_t g = start; g < end; g++) { d.gate_group_start[g] = start; d.gate_group_end[g] = end; }
    }
    d.quotient_degree_factor = r.usize32();
    const uint64_t num_gate_constraints = r.u64();      // implied by the gate list: checked below, once the constant columns are known
    d.num_constants = r.usize32();
    d.num_public_inputs = r.usize32();
    const uint64_t nk = r.u64();
    GL_REQUIRE(r.ok && nk == d.num_routed_wires && nk <= 80, GL_ERR_UNSUPPORTED, "k_is: one coset shift per routed wire, at most 80");
    for (uint64_t j = 0; j < nk; j++) d.k_is[j] = r.u64();
    const uint64_t num_partial_products = r.u64();      // implied: checked below
    const uint64_t nlp = r.u64(), nls = r.u64(), nluts = r.u64();
    GL_REQUIRE(!r.wide, GL_ERR_UNSUPPORTED, "a size field of CommonCircuitData exceeds 32 bits");
    GL_REQUIRE(r.ok, GL_ERR_ARG, "truncated CommonCircuitData");
    GL_REQUIRE(nluts <= GL_MAX_LUTS && ((nluts == 0 && nlp == 0 && nls == 0) || (nluts >= 1 && nlp == 7 && nls == glhost::LU_SEL_START_END + nluts)), GL_ERR_UNSUPPORTED,
               "lookup argument: at most 4 tables, 7 lookup polynomials and 4 + #tables lookup selectors per challenge");
    d.num_luts = (uint32_t)nluts; d.num_lookup_polys = (uint32_t)nlp; d.num_lookup_selectors = (uint32_t)nls;
    uint32_t total = 0;
    for (uint64_t t = 0; t < nluts; t++) {
        const uint64_t len = r.u64();
        GL_REQUIRE(r.ok && len >= 1 && total + len <= GL_MAX_LUT_ENTRIES, GL_ERR_UNSUPPORTED, "lookup tables: 1 .. 1024 entries together");
        d.lut_len[t] = (uint32_t)len;
        for (uint64_t k = 0; k < 2 * len && r.ok; k++) d.lut[2 * (size_t)total + k] = r.u16();
        total += (uint32_t)len;
    }
    GL_REQUIRE(r.ok, GL_ERR_ARG, "truncated CommonCircuitData");
    bool has_table_gate[GL_MAX_LUTS] = {false};
    for (uint64_t g = 0; g < ngates; g++) {
        if (d.gate_types[g] != 6 && d.gate_types[g] != 7) continue;
        unsigned t = 0;                 // the tables of `luts` are distinct (circuit_builder.rs is_stored): the first equal one is the gate's
        for (; t < nluts; t++) {
            const uint16_t* e = d.lut + 2 * (size_t)glhost::lut_offset(d, t);
            if (gate_table[g].size() == 2 * (size_t)d.lut_len[t] && std::equal(gate_table[g].begin(), gate_table[g].end(), e)) break;
        }
        GL_REQUIRE(t < nluts, GL_ERR_ARG, "a lookup gate's table is not one of CommonCircuitData's luts");
        d.gate_params[g] = (uint8_t)t;
        if (d.gate_types[g] == 7) {
            // LookupWire is prover data (circuit_data.rs:296-299), not part of these bytes: last_lut_row is the LookupTableGate's field,
            // first_lut_row follows from the table length; last_lu_row is unknown here and left 0 (the verifier does not need it; build()
            // reads it from the lookup selector columns)
            d.last_lut_row[t] = gate_last_lut_row[g];
            d.first_lut_row[t] = gate_last_lut_row[g] + glhost::lut_rows(d, t) - 1;
            has_table_gate[t] = true;
        }
    }
    for (uint64_t t = 0; t < nluts; t++) GL_REQUIRE(has_table_gate[t], GL_ERR_ARG, "a lookup table without its LookupTableGate");
    GL_REQUIRE(max_qdf == d.quotient_degree_factor && cfg_consts + d.num_selectors + d.num_lookup_selectors == d.num_constants, GL_ERR_UNSUPPORTED, "constants / quotient degree layout");
    GL_REQUIRE((!arith_ops || arith_ops == d.num_routed_wires / 4) && (!const_consts || const_consts == cfg_consts), GL_ERR_UNSUPPORTED, "gate parameters");
    {   // the two redundant fields must say what the rest implies (the writer derives them the same way)
        uint64_t max_constraints = 0;
        for (uint32_t g = 0; g < d.num_gates; g++) max_constraints = std::max<uint64_t>(max_constraints, gate_constraints(d.gate_types[g], d.gate_params[g], d));
        GL_REQUIRE(d.quotient_degree_factor >= 1 && num_gate_constraints == max_constraints &&
                   num_partial_products == (d.num_routed_wires + d.quotient_degree_factor - 1) / d.quotient_degree_factor - 1, GL_ERR_ARG,
                   "num_gate_constraints / num_partial_products differ from what the gates and the routed wires imply");
    }
    *out = d;
    if (consumed) *consumed = r.pos;
    return GL_OK;
}

// VerifierOnlyCircuitData (mod.rs:1889-1906): usize cap height, the cap's 2^height digests, the circuit digest
extern "C" int gl_verifier_only_to_bytes(uint32_t cap_height, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4], uint8_t* h_out, size_t cap,
                                         size_t* num_bytes) {
    GL_REQUIRE(constants_sigmas_cap && circuit_digest && num_bytes && cap_height <= 16, GL_ERR_ARG, "gl_verifier_only_to_bytes: bad argument");
    Writer w;
    w.u64(cap_height);
    for (size_t i = 0; i < (size_t(4) << cap_height); i++) w.field(constants_sigmas_cap[i]);
    for (int i = 0; i < 4; i++) w.field(circuit_digest[i]);
    *num_bytes = w.b.size();
    if (!h_out) return GL_OK;
    GL_REQUIRE(cap >= w.b.size(), GL_ERR_ARG, "gl_verifier_only_to_bytes: output too small");
    memcpy(h_out, w.b.data(), w.b.size());
    return GL_OK;
}
extern "C" int gl_verifier_only_from_bytes(const uint8_t* h_bytes, size_t num_bytes, uint32_t* cap_height, uint64_t* h_cap, size_t cap_words, uint64_t circuit_digest[4],
                                           size_t* consumed) {
    GL_REQUIRE(h_bytes && cap_height && circuit_digest, GL_ERR_ARG, "gl_verifier_only_from_bytes: null argument");
    Reader r(h_bytes, num_bytes);
    const uint64_t h = r.u64();
    GL_REQUIRE(r.ok && h <= 16, GL_ERR_ARG, "bad cap height");
    *cap_height = (uint32_t)h;
    const size_t words = size_t(4) << h;
    GL_REQUIRE(!h_cap || cap_words >= words, GL_ERR_ARG, "gl_verifier_only_from_bytes: cap buffer too small");
    for (size_t i = 0; i < words; i++) { const uint64_t v = r.u64(); if (h_cap) h_cap[i] = v; }
    for (int i = 0; i < 4; i++) circuit_digest[i] = r.u64();
    GL_REQUIRE(r.ok, GL_ERR_ARG, "truncated VerifierOnlyCircuitData");
    if (consumed) *consumed = r.pos;
    return GL_OK;
}

// VerifierCircuitData::from_bytes(..).verify(proof) (circuit_data.rs:208-238; the bytes are verifier_only || common,
// mod.rs:1908-1919)
extern "C" int gl_verify_bytes(const uint8_t* h_verifier_data, size_t num_data_bytes, const uint8_t* proof_bytes, size_t num_proof_bytes) {
    GL_REQUIRE(h_verifier_data && proof_bytes, GL_ERR_ARG, "gl_verify_bytes: null argument");
    uint32_t cap_height = 0; uint64_t digest[4]; size_t used = 0, used2 = 0;
    std::vector<uint64_t> cap(size_t(4) << 16);
    GL_TRY(gl_verifier_only_from_bytes(h_verifier_data, num_data_bytes, &cap_height, cap.data(), cap.size(), digest, &used));
    gl_circuit_desc d;
    GL_TRY(gl_common_data_from_bytes(h_verifier_data + used, num_data_bytes - used, &d, &used2));
    GL_REQUIRE(used + used2 == num_data_bytes, GL_ERR_ARG, "trailing bytes after VerifierCircuitData");
    GL_REQUIRE(cap_height == d.cap_height, GL_ERR_ARG, "cap height of the verifier data differs from the FRI configuration");
    return gl_verify(&d, cap.data(), digest, proof_bytes, num_proof_bytes);
}
