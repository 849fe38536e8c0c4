// ORACLE (test infrastructure only): flat C entry points over the CPU restatement so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
// The product library (libplonky2_mi355x.so) never links or loads this file.
#include "gl_prover.hpp"
#include <cstring>

using namespace orc;

extern "C" {

// op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 inverse(a), 5 canon(a), 6 mul_add(a + b*c)
void orc_field_op(int op, const u64* a, const u64* b, const u64* c, u64* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        u64 r = 0;
        switch (op) {
            case 0: r = add(a[i], b[i]); break;
            case 1: r = sub(a[i], b[i]); break;
            case 2: r = mul(a[i], b[i]); break;
            case 3: r = neg(a[i]); break;
            case 4: r = inv(a[i]); break;
            case 5: r = a[i]; break;
            case 6: r = mul_add(a[i], b[i], c[i]); break;
        }
        out[i] = canon(r);
    }
}
// op: 0 add, 1 sub, 2 mul, 3 inverse(a); interleaved (a0,a1) pairs
void orc_ext_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Ext2 x{a[2 * i], a[2 * i + 1]}, y{0, 0}, r{0, 0};
        if (b) y = Ext2{b[2 * i], b[2 * i + 1]};
        switch (op) {
            case 0: r = eadd(x, y); break;
            case 1: r = esub(x, y); break;
            case 2: r = emul(x, y); break;
            case 3: r = einv(x); break;
        }
        out[2 * i] = canon(r.a); out[2 * i + 1] = canon(r.b);
    }
}
u64 orc_primitive_root(unsigned n_log) { return canon(primitive_root_of_unity(n_log)); }
u64 orc_inverse_2exp(unsigned e) { return canon(inverse_2exp(e)); }

static void canon_all(std::vector<u64>& v) { for (auto& x : v) x = canon(x); }

// batch of `batch` contiguous polynomials of length n, transformed in place; outputs canonical
void orc_fft(u64* data, size_t n, size_t batch, unsigned zero_factor) {
    RootTable rt = fft_root_table(n);
    for (size_t b = 0; b < batch; b++) {
        std::vector<u64> v(data + b * n, data + (b + 1) * n);
        fft_inplace(v, zero_factor, &rt); canon_all(v);
        memcpy(data + b * n, v.data(), n * 8);
    }
}
void orc_ifft(u64* data, size_t n, size_t batch) {
    RootTable rt = fft_root_table(n);
    for (size_t b = 0; b < batch; b++) {
        std::vector<u64> v(data + b * n, data + (b + 1) * n);
        ifft_inplace(v, &rt); canon_all(v);
        memcpy(data + b * n, v.data(), n * 8);
    }
}
// forward (inverse = 0) or inverse transforms of `batch` polynomials, parallel over polynomials like the
// reference's par_iter over columns (plonky2/src/fri/oracle.rs:54,111-118); each FFT is single-threaded.
void orc_fft_mt(u64* data, size_t n, size_t batch, int inverse, unsigned threads) {
    RootTable rt = fft_root_table(n);
    parallel_for(batch, threads, [&](size_t b) {
        std::vector<u64> v(data + b * n, data + (b + 1) * n);
        if (inverse) ifft_inplace(v, &rt); else fft_inplace(v, 0, &rt);
        canon_all(v);
        memcpy(data + b * n, v.data(), n * 8);
    });
}
void orc_coset_fft(u64* data, size_t n, size_t batch, u64 shift, unsigned zero_factor) {
    RootTable rt = fft_root_table(n);
    for (size_t b = 0; b < batch; b++) {
        std::vector<u64> v(data + b * n, data + (b + 1) * n);
        coset_fft_inplace(v, shift, zero_factor, &rt); canon_all(v);
        memcpy(data + b * n, v.data(), n * 8);
    }
}
void orc_coset_ifft(u64* data, size_t n, size_t batch, u64 shift) {
    RootTable rt = fft_root_table(n);
    for (size_t b = 0; b < batch; b++) {
        std::vector<u64> v(data + b * n, data + (b + 1) * n);
        coset_ifft_inplace(v, shift, &rt); canon_all(v);
        memcpy(data + b * n, v.data(), n * 8);
    }
}
// coeffs [batch][n] -> values on 7*H_{n<<rate_bits}, [batch][n<<rate_bits], natural order
void orc_lde(const u64* coeffs, size_t n, size_t batch, unsigned rate_bits, u64* out, unsigned threads) {
    size_t N = n << rate_bits;
    RootTable rt = fft_root_table(N);
    parallel_for(batch, threads, [&](size_t b) {
        std::vector<u64> c(coeffs + b * n, coeffs + (b + 1) * n);
        std::vector<u64> v = lde_coset(c, rate_bits, GL_GENERATOR, &rt);
        for (size_t i = 0; i < N; i++) out[b * N + i] = canon(v[i]);
    });
}
void orc_evaluate_naive(const u64* coeffs, size_t n, u64* out) {
    std::vector<u64> c(coeffs, coeffs + n);
    std::vector<u64> v = evaluate_naive(c);
    for (size_t i = 0; i < n; i++) out[i] = canon(v[i]);
}

void orc_poseidon(u64* state, size_t count, int naive) {
    for (size_t k = 0; k < count; k++) {
        PState s; for (int i = 0; i < 12; i++) s[i] = state[12 * k + i];
        if (naive) poseidon_naive(s); else poseidon(s);
        for (int i = 0; i < 12; i++) state[12 * k + i] = canon(s[i]);
    }
}
void orc_hash_or_noop(const u64* in, size_t n, u64* out4) {
    Digest d = hash_or_noop(in, n);
    for (int i = 0; i < 4; i++) out4[i] = canon(d.e[i]);
}
void orc_hash_no_pad(const u64* in, size_t n, u64* out4) {
    Digest d = hash_no_pad(in, n);
    for (int i = 0; i < 4; i++) out4[i] = canon(d.e[i]);
}
void orc_two_to_one(const u64* l, const u64* r, u64* out4) {
    Digest a, b; memcpy(a.e, l, 32); memcpy(b.e, r, 32);
    Digest d = two_to_one(a, b);
    for (int i = 0; i < 4; i++) out4[i] = canon(d.e[i]);
}

// ---- Merkle tree handle -------------------------------------------------------------------------
void* orc_merkle_new(const u64* leaves, size_t num_leaves, size_t leaf_len, unsigned cap_height) {
    std::vector<u64> l(leaves, leaves + num_leaves * leaf_len);
    return new MerkleTree(merkle_build(std::move(l), num_leaves, leaf_len, cap_height));
}
void orc_merkle_free(void* t) { delete (MerkleTree*)t; }
static void write_digests(const std::vector<Digest>& v, u64* out) {
    for (size_t i = 0; i < v.size(); i++) for (int k = 0; k < 4; k++) out[4 * i + k] = canon(v[i].e[k]);
}
void orc_merkle_cap(const void* t, u64* out) { write_digests(((const MerkleTree*)t)->cap(), out); }
size_t orc_merkle_prove(const void* t, size_t leaf_index, u64* out) {
    auto s = ((const MerkleTree*)t)->prove(leaf_index);
    write_digests(s, out);
    return s.size();
}
size_t orc_merkle_num_levels(const void* t) { return ((const MerkleTree*)t)->levels.size(); }
void orc_merkle_level(const void* t, size_t level, u64* out) { write_digests(((const MerkleTree*)t)->levels[level], out); }
int orc_merkle_verify(const u64* leaf, size_t leaf_len, size_t leaf_index, const u64* cap, size_t cap_len,
                      const u64* siblings, size_t nsib) {
    std::vector<Digest> c(cap_len), s(nsib);
    memcpy(c.data(), cap, cap_len * 32); memcpy(s.data(), siblings, nsib * 32);
    return merkle_verify(leaf, leaf_len, leaf_index, c, s) ? 1 : 0;
}

// ---- PolynomialBatch handle ---------------------------------------------------------------------
// cols: [ncols][n] contiguous
void* orc_batch_new(const u64* cols, size_t ncols, size_t n, unsigned rate_bits, unsigned cap_height,
                    int from_values, unsigned threads) {
    std::vector<std::vector<u64>> v(ncols);
    for (size_t c = 0; c < ncols; c++) v[c].assign(cols + c * n, cols + (c + 1) * n);
    PolynomialBatch* b = new PolynomialBatch(from_values ? batch_from_values(std::move(v), rate_bits, cap_height, threads)
                                                         : batch_from_coeffs(std::move(v), rate_bits, cap_height, threads));
    return b;
}
void orc_batch_free(void* b) { delete (PolynomialBatch*)b; }
void orc_batch_cap(const void* b, u64* out) { write_digests(((const PolynomialBatch*)b)->tree.cap(), out); }
void orc_batch_coeffs(const void* b, u64* out) {
    const PolynomialBatch* pb = (const PolynomialBatch*)b;
    size_t n = pb->polynomials[0].size();
    for (size_t c = 0; c < pb->ncols(); c++) for (size_t i = 0; i < n; i++) out[c * n + i] = canon(pb->polynomials[c][i]);
}
// leaves in Merkle order (row j = LDE point bitrev(j)), row-major N x ncols
void orc_batch_leaves(const void* b, u64* out) {
    const PolynomialBatch* pb = (const PolynomialBatch*)b;
    for (size_t i = 0; i < pb->tree.leaves.size(); i++) out[i] = canon(pb->tree.leaves[i]);
}
void orc_batch_leaf(const void* b, size_t index, u64* out) {
    const PolynomialBatch* pb = (const PolynomialBatch*)b;
    for (size_t c = 0; c < pb->ncols(); c++) out[c] = canon(pb->tree.leaf(index)[c]);
}
size_t orc_batch_prove(const void* b, size_t leaf_index, u64* out) {
    return orc_merkle_prove(&((const PolynomialBatch*)b)->tree, leaf_index, out);
}
size_t orc_batch_num_levels(const void* b) { return ((const PolynomialBatch*)b)->tree.levels.size(); }
void orc_batch_level(const void* b, size_t level, u64* out) { write_digests(((const PolynomialBatch*)b)->tree.levels[level], out); }


// ---- matmul circuit / witness / prover / verifier ------------------------------------------------------
struct OrcProof { Proof proof; ProverTrace trace; std::vector<uint8_t> bytes; };

// util/partial_products.rs helpers, for the reference's own test vector (partial_products.rs:114-146)
size_t orc_quotient_chunk_products(const u64* v, size_t n, size_t max_degree, u64* out) {
    auto r = quotient_chunk_products(std::vector<u64>(v, v + n), max_degree);
    for (size_t i = 0; i < r.size(); i++) out[i] = canon(r[i]);
    return r.size();
}
void orc_partial_products_and_z_gx(u64 z_x, const u64* chunks, size_t n, u64* out) {
    auto r = partial_products_and_z_gx(z_x, std::vector<u64>(chunks, chunks + n));
    for (size_t i = 0; i < r.size(); i++) out[i] = canon(r[i]);
}
// check_partial_products over the base field: out gets ceil(n / max_degree) constraint values
size_t orc_check_partial_products(const u64* num, const u64* den, size_t n, const u64* partials, size_t nparts, u64 z_x, u64 z_gx, size_t max_degree, u64* out) {
    std::vector<u64> res;
    check_partial_products<u64>(num, den, n, partials, nparts, z_x, z_gx, max_degree, res);
    for (size_t i = 0; i < res.size(); i++) out[i] = canon(res[i]);
    return res.size();
}
// ---- stand-alone pieces of prove_openings, for the reference's own property tests (util/reducing.rs, polynomial/division.rs, cosets.rs)
// out[2 n]: sum_j alpha^j polys[j] (polys: npolys x n base-field coefficients, row-major)
void orc_reduce_polys_base(const u64* alpha2, const u64* polys, size_t npolys, size_t n, u64* out) {
    std::vector<std::vector<u64>> ps(npolys);
    std::vector<const std::vector<u64>*> ptr;
    for (size_t j = 0; j < npolys; j++) { ps[j].assign(polys + j * n, polys + (j + 1) * n); ptr.push_back(&ps[j]); }
    auto r = reduce_polys_base(Ext2{alpha2[0], alpha2[1]}, ptr, n);
    for (size_t i = 0; i < n; i++) { out[2 * i] = canon(r[i].a); out[2 * i + 1] = canon(r[i].b); }
}
// out[2]: ReducingFactor::reduce of n extension values
void orc_reduce_ext(const u64* alpha2, const u64* vals, size_t n, u64* out2) {
    std::vector<Ext2> v(n);
    for (size_t i = 0; i < n; i++) v[i] = Ext2{vals[2 * i], vals[2 * i + 1]};
    Ext2 r = reduce_ext(Ext2{alpha2[0], alpha2[1]}, v);
    out2[0] = canon(r.a); out2[1] = canon(r.b);
}
// out[2 (n - 1)]: (p(X) - p(z)) / (X - z) for n extension coefficients
void orc_divide_by_linear(const u64* coeffs, size_t n, const u64* z2, u64* out) {
    std::vector<Ext2> c(n);
    for (size_t i = 0; i < n; i++) c[i] = Ext2{coeffs[2 * i], coeffs[2 * i + 1]};
    auto q = divide_by_linear(c, Ext2{z2[0], z2[1]});
    for (size_t i = 0; i < q.size(); i++) { out[2 * i] = canon(q[i].a); out[2 * i + 1] = canon(q[i].b); }
}
// get_unique_coset_shifts (field/src/cosets.rs:9-24): g^0 .. g^(num - 1), g = MULTIPLICATIVE_GROUP_GENERATOR (the k_is of build())
void orc_unique_coset_shifts(size_t num, u64* out) { u64 x = 1; for (size_t i = 0; i < num; i++) { out[i] = canon(x); x = mul(x, GL_GENERATOR); } }

void* orc_circuit_new(size_t m, unsigned threads) { return new CircuitData(build_matmul_circuit(m, threads)); }
// CommonCircuitData only (what verify() needs besides the cap and the digest): cheap even for m = 128
void* orc_circuit_new_verifier_only(size_t m) { return new CircuitData(build_matmul_circuit(m, 1, false)); }
// test circuits over the same gate set (gl_circuit.hpp build_test_circuit): kind 1 hash-only, kind 2 arithmetic chain without PIs
void* orc_circuit_new_kind(int kind, size_t param, unsigned threads) { return new CircuitData(build_test_circuit(kind, param, threads)); }
// what a gl_circuit_desc needs beyond orc_circuit_info: per gate (selector index, group start, group end), k_is, FRI arities
size_t orc_circuit_selector_groups(const void* c, u64* out /* [num_gates][3] */) {
    const CircuitData* cd = (const CircuitData*)c; const SelectorsInfo& si = cd->common.selectors;
    for (size_t i = 0; i < si.gates.size(); i++) { size_t g = si.selector_indices[i]; out[3 * i] = g; out[3 * i + 1] = si.groups[g].first; out[3 * i + 2] = si.groups[g].second; }
    return si.gates.size();
}
void orc_circuit_k_is(const void* c, u64* out80) { const CircuitData* cd = (const CircuitData*)c; for (size_t i = 0; i < cd->common.k_is.size(); i++) out80[i] = canon(cd->common.k_is[i]); }
size_t orc_circuit_num_inputs(const void* c, size_t* nb) { const CircuitData* cd = (const CircuitData*)c; if (nb) *nb = cd->b_targets.size(); return cd->a_targets.size(); }
// verify ProofWithPublicInputs bytes against VerifierOnlyCircuitData = (constants_sigmas_cap[2^cap_height][4], circuit_digest[4])
// 0 = accepted, 1 = rejected, 2 = malformed bytes
static thread_local const char* g_verify_msg2 = "";
int orc_verify_bytes(const void* c, const u64* cap, const u64* digest, const uint8_t* bytes, size_t len) {
    const CircuitData* cd = (const CircuitData*)c;
    Proof p;
    if (!proof_from_bytes(cd->common, bytes, len, p)) { g_verify_msg2 = "malformed proof bytes"; return 2; }
    std::vector<Digest> capv(size_t(1) << cd->common.config.cap_height);
    memcpy(capv.data(), cap, capv.size() * 32);
    Digest dg; memcpy(dg.e, digest, 32);
    const char* m = verify(cd->common, capv, dg, p);
    g_verify_msg2 = m ? m : "";
    return m ? 1 : 0;
}
const char* orc_verify_bytes_message(void) { return g_verify_msg2; }

// ---- CommonCircuitData::to_bytes / VerifierCircuitData::to_bytes restated from the oracle's own CommonData --------------------
// (util/serialization/mod.rs:1596-1790,1889-1919; gate tags gate_serialization.rs:89-107; gates/arithmetic_base.rs:63-65,
// gates/constant.rs:49-51).  A second, independent writer: tests demand that the product's bytes are identical.
namespace {
struct ByteSink {
    std::vector<uint8_t> b;
    void usize(u64 x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void word32(uint32_t x) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void byte(uint8_t x) { b.push_back(x); }
};
static void sink_fri_config(ByteSink& o, const CircuitConfig& c) {
    o.usize(c.rate_bits); o.usize(c.cap_height); o.usize(c.num_query_rounds); o.word32(c.proof_of_work_bits);
    o.byte(1); o.usize(c.fri_arity_bits); o.usize(c.fri_final_poly_bits);                  // ConstantArityBits
}
static void sink_lut(ByteSink& o, const std::vector<std::pair<uint16_t, uint16_t>>& lut) {      // write_lut (mod.rs:2077-2085)
    o.usize(lut.size());
    for (auto& e : lut) { o.byte((uint8_t)e.first); o.byte((uint8_t)(e.first >> 8)); o.byte((uint8_t)e.second); o.byte((uint8_t)(e.second >> 8)); }
}
static void sink_common(ByteSink& o, const CommonData& cm) {
    const CircuitConfig& c = cm.config;
    o.usize(c.num_wires); o.usize(c.num_routed_wires); o.usize(c.num_constants); o.usize(100 /* security_bits, circuit_data.rs:77 */); o.usize(c.num_challenges);
    o.usize(c.max_quotient_degree_factor); o.byte(1); o.byte(0);
    sink_fri_config(o, c);
    sink_fri_config(o, c);                                                                 // FriParams.config
    o.usize(cm.fri_reduction_arity_bits.size()); for (auto a : cm.fri_reduction_arity_bits) o.usize(a);
    o.usize(cm.degree_bits); o.byte(0);
    o.usize(cm.selectors.gates.size());
    for (size_t gi = 0; gi < cm.selectors.gates.size(); gi++) {
        const GateType g = cm.selectors.gates[gi];
        const size_t li = cm.selectors.gate_params.empty() ? 0 : cm.selectors.gate_params[gi];
        switch (g) {
            case GATE_ARITHMETIC: o.word32(0); o.usize(c.num_routed_wires / 4); break;     // ArithmeticGate::new_from_config: num_ops = routed / 4
            case GATE_CONSTANT: o.word32(3); o.usize(c.num_constants); break;
            case GATE_NOOP: o.word32(9); break;
            case GATE_POSEIDON: o.word32(11); break;
            case GATE_RANDOM_ACCESS: { RandomAccess ra(li); o.word32(13); o.usize(ra.bits); o.usize(ra.num_copies); o.usize(ra.num_extra_constants); break; }   // random_access.rs:123-128
            case GATE_EXPONENTIATION: o.word32(5); o.usize(EXP_POWER_BITS); break;        // write_usize(num_power_bits) (exponentiation.rs:79-81)
            case GATE_BASE_SUM: o.word32(2); o.usize(BASE_SUM_LIMBS); break;              // BaseSumGate<2>: write_usize(num_limbs) (base_sum.rs:53-55)
            case GATE_LOOKUP: o.word32(6); o.usize(LOOKUP_SLOTS); sink_lut(o, cm.luts[li]); break;                    // lookup.rs:59-62
            case GATE_LOOKUP_TABLE: o.word32(7); o.usize(LOOKUP_TABLE_SLOTS); sink_lut(o, cm.luts[li]); o.usize(cm.last_lut_rows[li]); break;   // lookup_table.rs:70-74
            default: o.word32(12); break;                                                  // PublicInputGate
        }
    }
    o.usize(cm.selectors.selector_indices.size()); for (auto i : cm.selectors.selector_indices) o.usize(i);
    o.usize(cm.selectors.groups.size()); for (auto& g : cm.selectors.groups) { o.usize(g.first); o.usize(g.second); }
    o.usize(cm.quotient_degree_factor); o.usize(cm.num_gate_constraints); o.usize(cm.num_constants); o.usize(cm.num_public_inputs);
    o.usize(cm.k_is.size()); for (auto k : cm.k_is) o.usize(canon(k));
    o.usize(cm.num_partial_products);
    o.usize(cm.num_lookup_polys); o.usize(cm.num_lookup_selectors); o.usize(cm.luts.size());       // mod.rs:1776-1782
    for (auto& lut : cm.luts) sink_lut(o, lut);
}
}
// kind 0: CommonCircuitData, kind 1: VerifierCircuitData (verifier_only || common).  Returns the size; writes when cap suffices.
size_t orc_circuit_data_bytes(const void* c, int kind, uint8_t* out, size_t cap) {
    const CircuitData* cd = (const CircuitData*)c;
    ByteSink o;
    if (kind == 1) {
        auto capv = cd->constants_sigmas_commitment.tree.cap();
        o.usize(cd->common.config.cap_height);
        for (auto& d : capv) for (int i = 0; i < 4; i++) o.usize(canon(d.e[i]));
        for (int i = 0; i < 4; i++) o.usize(canon(cd->circuit_digest.e[i]));
    }
    sink_common(o, cd->common);
    if (out && cap >= o.b.size()) memcpy(out, o.b.data(), o.b.size());
    return o.b.size();
}
void orc_circuit_free(void* c) { delete (CircuitData*)c; }
// out: [degree_bits, num_constants, num_gate_constraints, num_partial_products, num_public_inputs, num_selectors,
//       num_fri_rounds, final_poly_len, pi_row, constant_row (the first ConstantGate), num_arith_ops, num_poseidon_rows]
void orc_circuit_info(const void* c, u64* out) {
    const CircuitData* cd = (const CircuitData*)c;
    const CommonData& cm = cd->common;
    u64 v[12] = {cm.degree_bits, cm.num_constants, cm.num_gate_constraints, cm.num_partial_products, cm.num_public_inputs,
                 cm.selectors.num_selectors(), cm.fri_reduction_arity_bits.size(), cm.final_poly_len(), cd->pi_row, cd->constant_wires.empty() ? 0 : cd->constant_wires[0].row,
                 cd->arith_ops.size(), cd->poseidon_rows.size()};
    memcpy(out, v, sizeof v);
}
// lookup data of a circuit: out3 = [num_lookup_polys, num_lookup_selectors, num_luts]; per table t: rows[4 t ..] = [last_lu_row, last_lut_row,
// first_lut_row, entries]; lut (may be null) receives the (input, output) pairs of all tables, one after the other; gate_params (may be null):
// per gate of the sorted gate list, the table of a LookupGate / LookupTableGate (else 0)
void orc_circuit_lookup_info(const void* c, u64* out3, u64* rows, uint16_t* lut, uint8_t* gate_params) {
    const CircuitData* cd = (const CircuitData*)c;
    const CommonData& cm = cd->common;
    out3[0] = cm.num_lookup_polys; out3[1] = cm.num_lookup_selectors; out3[2] = cm.luts.size();
    size_t off = 0;
    for (size_t t = 0; t < cm.luts.size(); t++) {
        rows[4 * t] = rows[4 * t + 2] = 0;
        rows[4 * t + 1] = cm.last_lut_rows[t];
        if (t < cd->lookup_rows.size()) { rows[4 * t] = cd->lookup_rows[t].last_lu_gate; rows[4 * t + 1] = cd->lookup_rows[t].last_lut_gate; rows[4 * t + 2] = cd->lookup_rows[t].first_lut_gate; }
        rows[4 * t + 3] = cm.luts[t].size();
        if (lut) for (auto& e : cm.luts[t]) { lut[2 * off] = e.first; lut[2 * off + 1] = e.second; off++; }
    }
    if (gate_params) for (size_t i = 0; i < cm.selectors.gates.size(); i++) gate_params[i] = (uint8_t)(cm.selectors.gate_params.empty() ? 0 : cm.selectors.gate_params[i]);
}
void orc_circuit_digest(const void* c, u64* out4) { for (int i = 0; i < 4; i++) out4[i] = canon(((const CircuitData*)c)->circuit_digest.e[i]); }
void orc_circuit_cs_cap(const void* c, u64* out) { write_digests(((const CircuitData*)c)->constants_sigmas_commitment.tree.cap(), out); }
void orc_circuit_constants_sigmas(const void* c, u64* out) {
    const CircuitData* cd = (const CircuitData*)c; size_t n = cd->common.degree();
    for (size_t k = 0; k < cd->constants_sigmas.size(); k++) for (size_t i = 0; i < n; i++) out[k * n + i] = canon(cd->constants_sigmas[k][i]);
}
void orc_circuit_row_gates(const void* c, uint8_t* out) {
    const CircuitData* cd = (const CircuitData*)c;
    for (size_t i = 0; i < cd->row_gate.size(); i++) out[i] = (uint8_t)cd->row_gate[i];
}
// gate order used by the selectors (sorted by (degree, id)), as GateType codes
size_t orc_circuit_gate_order(const void* c, uint8_t* out) {
    const CircuitData* cd = (const CircuitData*)c;
    for (size_t i = 0; i < cd->common.selectors.gates.size(); i++) out[i] = (uint8_t)cd->common.selectors.gates[i];
    return cd->common.selectors.gates.size();
}
void* orc_witness_new(const void* c, const u64* a, const u64* b, u64 filler_seed) {
    const CircuitData* cd = (const CircuitData*)c; const size_t na = cd->a_targets.size(), nb = cd->b_targets.size();
    return new Witness(generate_witness(*cd, std::vector<u64>(a, a + na), std::vector<u64>(b, b + nb), filler_seed));
}
void orc_witness_free(void* w) { delete (Witness*)w; }
void orc_witness_wires(const void* w, u64* out) {
    const Witness* wt = (const Witness*)w; size_t n = wt->wire_values[0].size();
    for (size_t j = 0; j < wt->wire_values.size(); j++) memcpy(out + j * n, wt->wire_values[j].data(), n * 8);
}
size_t orc_witness_public_inputs(const void* w, u64* out) {
    const Witness* wt = (const Witness*)w;
    if (out) memcpy(out, wt->public_inputs.data(), wt->public_inputs.size() * 8);
    return wt->public_inputs.size();
}
// a witness handle from an externally supplied wire matrix [135][n] + public inputs
void* orc_witness_from_matrix(const u64* wires, size_t num_wires, size_t n, const u64* pis, size_t npis) {
    Witness* w = new Witness();
    w->wire_values.resize(num_wires);
    for (size_t j = 0; j < num_wires; j++) w->wire_values[j].assign(wires + j * n, wires + (j + 1) * n);
    w->public_inputs.assign(pis, pis + npis);
    return w;
}
void* orc_prove(const void* c, const void* w, unsigned threads) {
    OrcProof* p = new OrcProof();
    if (!prove(*(const CircuitData*)c, *(const Witness*)w, threads, p->proof, &p->trace)) { delete p; return nullptr; }
    p->bytes = proof_to_bytes(p->proof);
    return p;
}
void orc_proof_free(void* p) { delete (OrcProof*)p; }
size_t orc_proof_bytes(const void* p, uint8_t* out, size_t cap) {
    const OrcProof* op = (const OrcProof*)p;
    if (out && cap >= op->bytes.size()) memcpy(out, op->bytes.data(), op->bytes.size());
    return op->bytes.size();
}
// out: betas[2] gammas[2] alphas[2] zeta[2] fri_alpha[2] pow_witness pi_hash[4] then fri_betas (2 each)
size_t orc_proof_challenges(const void* p, u64* out) {
    const OrcProof* op = (const OrcProof*)p; size_t k = 0;
    for (u64 v : op->trace.betas) out[k++] = canon(v);
    for (u64 v : op->trace.gammas) out[k++] = canon(v);
    for (u64 v : op->trace.alphas) out[k++] = canon(v);
    out[k++] = canon(op->trace.zeta.a); out[k++] = canon(op->trace.zeta.b);
    out[k++] = canon(op->trace.fri_alpha.a); out[k++] = canon(op->trace.fri_alpha.b);
    out[k++] = canon(op->proof.opening_proof.pow_witness);
    for (int i = 0; i < 4; i++) out[k++] = canon(op->trace.public_inputs_hash.e[i]);
    for (auto& b : op->trace.fri_betas) { out[k++] = canon(b.a); out[k++] = canon(b.b); }
    return k;
}
void orc_proof_zs_partial_products(const void* p, u64* out) {
    const OrcProof* op = (const OrcProof*)p; size_t n = op->trace.zs_partial_products[0].size();
    for (size_t c = 0; c < op->trace.zs_partial_products.size(); c++) for (size_t i = 0; i < n; i++) out[c * n + i] = canon(op->trace.zs_partial_products[c][i]);
}
void orc_proof_quotient_chunks(const void* p, u64* out) {
    const OrcProof* op = (const OrcProof*)p; size_t n = op->trace.quotient_chunks[0].size();
    for (size_t c = 0; c < op->trace.quotient_chunks.size(); c++) for (size_t i = 0; i < n; i++) out[c * n + i] = canon(op->trace.quotient_chunks[c][i]);
}
void orc_proof_caps(const void* p, u64* out) {          // wires, zs_pp, quotient: 3 x 16 x 4
    const OrcProof* op = (const OrcProof*)p;
    write_digests(op->proof.wires_cap, out); write_digests(op->proof.zs_pp_cap, out + 64); write_digests(op->proof.quotient_cap, out + 128);
}
size_t orc_proof_query_indices(const void* p, u64* out) {
    const OrcProof* op = (const OrcProof*)p;
    for (size_t i = 0; i < op->trace.query_indices.size(); i++) out[i] = op->trace.query_indices[i];
    return op->trace.query_indices.size();
}
void orc_proof_final_poly_initial(const void* p, u64* out) {
    const OrcProof* op = (const OrcProof*)p;
    for (size_t i = 0; i < op->trace.final_poly_coeffs_initial.size(); i++) { out[2 * i] = canon(op->trace.final_poly_coeffs_initial[i].a); out[2 * i + 1] = canon(op->trace.final_poly_coeffs_initial[i].b); }
}
static thread_local const char* g_verify_msg = "";
// 0 = accepted
int orc_verify(const void* c, const void* p) {
    const char* m = verify(*(const CircuitData*)c, ((const OrcProof*)p)->proof);
    g_verify_msg = m ? m : "";
    return m ? 1 : 0;
}
const char* orc_verify_message(void) { return g_verify_msg; }
// flips one byte-level component of the proof (test helper for verifier soundness smoke checks)
void orc_proof_tamper(void* p, int what) {
    OrcProof* op = (OrcProof*)p;
    switch (what) {
        case 0: op->proof.openings.wires[3].a = add(op->proof.openings.wires[3].a, 1); break;
        case 1: op->proof.opening_proof.pow_witness += 1; break;
        case 2: op->proof.opening_proof.final_poly[0].b = add(op->proof.opening_proof.final_poly[0].b, 1); break;
        case 3: op->proof.public_inputs[0] = add(op->proof.public_inputs[0], 1); break;
        case 4: op->proof.opening_proof.query_round_proofs[5].initial[1].first[7] = add(op->proof.opening_proof.query_round_proofs[5].initial[1].first[7], 1); break;
        case 5: op->proof.quotient_cap[2].e[1] = add(op->proof.quotient_cap[2].e[1], 1); break;
    }
    op->bytes = proof_to_bytes(op->proof);      // the serialised form follows the tampered struct
}

}  // extern "C"
