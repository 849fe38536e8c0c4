/*
 * plonky2_mi355x.h -- C ABI of libplonky2_mi355x.so, the MI355X (gfx950) backend for the prove() hot path
 * of the Plonky2 matrix-multiplication demo circuit.
 *
 * The reference (Lain-Iwakuro/Plonky2-Demo, a plonky2 fork) has no FFI seam; this header cuts one at the
 * `PolynomialBatch` type (plonky2/src/fri/oracle.rs:30-37) and at the prover-side consumers that reach
 * into it (plonky2/src/plonk/prover.rs:102-329).  Every entry point names the reference interface it
 * replaces.  INTEGRATION.md shows the Rust `extern "C"` block and the patch to the call sites.
 *
 * Conventions
 *  - A field element is a little-endian uint64_t, layout-compatible with `#[repr(transparent)]
 *    GoldilocksField(pub u64)` (field/src/goldilocks_field.rs:23-25).  Inputs may be non-canonical
 *    (any u64); every output is canonical (< p = 2^64 - 2^32 + 1).
 *  - An extension element (F_p[X]/(X^2-7)) is two consecutive uint64_t: (a0, a1)
 *    (field/src/extension/quadratic.rs:14).  A digest (HashOut) is four uint64_t
 *    (plonky2/src/hash/hash_types.rs:20-24).
 *  - `h_` pointers are host memory owned by the caller; `d_` pointers are device (HIP) memory on the
 *    context's device.  Opaque handles own device memory and are released with their *_free.
 *  - All functions return GL_OK (0) or a GL_ERR_* code and never unwind; gl_last_error() gives the
 *    text for the calling thread.  The reference panics on shape errors (oracle.rs:114,
 *    merkle_tree.rs:137-143, fft.rs:175-181); here they are GL_ERR_ARG.
 *  - A gl_ctx is bound to one device and one HIP stream; calls on one ctx are issued in order on that
 *    stream.  Different ctxs may be used concurrently from different threads; ONE ctx runs one computing call at a
 *    time.  The plain copies out of a handle -- gl_batch_cap / gl_batch_coeffs / gl_batch_lde, gl_merkle_cap,
 *    gl_circuit_digest / gl_circuit_constants_sigmas_cap -- go through the context the handle was created on and MAY be
 *    called from any thread while that context is computing on another (a verifier thread reads the cap of a circuit that
 *    is proving); the getters that gather on the device first (get_leaf, prove, get_lde_values) count as computing calls.
 *  - Lifetime: every handle created on a context (gl_batch, gl_merkle, gl_circuit, gl_fri, gl_matmul_witgen) holds a
 *    reference to it.  gl_ctx_destroy() drops the creator's reference: after it the gl_ctx pointer must not be passed to
 *    any entry point again, but handles created earlier stay valid (they keep the stream, tables and allocator alive)
 *    and may be used and freed afterwards IN ANY ORDER; the context is torn down when the last of them is freed.  A
 *    binding may therefore give every handle a plain destructor (Rust `Drop`) without ordering them.  A gl_batch passed
 *    to gl_fri_combine, and the gl_host_circuit passed to gl_matmul_witgen_create / gl_prover_pool_create, are
 *    borrowed and must outlive the borrower.  Memory from gl_dev_alloc is not tied to the context (gl_dev_free accepts a
 *    null ctx).
 *  - There is NO CPU fallback: without a HIP device every compute entry point fails with GL_ERR_HIP.
 */
#ifndef PLONKY2_MI355X_H
#define PLONKY2_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL_OK 0
#define GL_ERR_ARG 1          /* bad shape / null pointer / unsupported size          */
#define GL_ERR_HIP 2          /* HIP runtime error (no device, OOM, launch failure)   */
#define GL_ERR_UNSUPPORTED 3  /* e.g. blinding != 0 (zero-knowledge salts)            */
#define GL_ERR_ZETA_IN_SUBGROUP 4 /* prover.rs:280-283: opening point lies in H       */
#define GL_ERR_INTERNAL 5
#define GL_ERR_VERIFY 6       /* gl_verify: the proof is rejected (gl_last_error says which check failed) */

typedef struct gl_ctx gl_ctx;
typedef struct gl_batch gl_batch;      /* device-resident PolynomialBatch (fri/oracle.rs:30-37)  */
typedef struct gl_merkle gl_merkle;    /* device-resident MerkleTree (hash/merkle_tree.rs:39-55) */
typedef struct gl_host_circuit gl_host_circuit;   /* host-only circuit description + witness recipe        */
typedef struct gl_circuit gl_circuit;  /* device-resident prover data (ProverOnlyCircuitData + CommonCircuitData) */
typedef struct gl_proof gl_proof;      /* ProofWithPublicInputs + the prover's intermediate values              */

/* What prove() needs from CommonCircuitData / ProverOnlyCircuitData (plonky2/src/plonk/circuit_data.rs:240-330,
 * 380-440) besides the constants/sigma value columns.  Gate types: 0 Noop, 1 Constant, 2 PublicInput,
 * 3 Arithmetic(20 ops), 4 Poseidon -- the gate set of the matmul demo circuit -- and 5 BaseSumGate<2> with the 63 limbs
 * of BaseSumGate::new_from_config (gates/base_sum.rs:31-35; range_check / split_le), 6 LookupGate and 7 LookupTableGate
 * (the lookup argument, several tables: fields at the end of this struct; phase API: the *_lookups variants), 8 ExponentiationGate
 * with the 66 power bits of new_from_config (gates/exponentiation.rs:43-53; CircuitBuilder::exp), 9 RandomAccessGate::new_from_config
 * (gates/random_access.rs:55-72; CircuitBuilder::random_access, Merkle caps) with its index bits (1..6) in `gate_params`; `gate_types` is the list
 * `common_data.gates` (sorted by degree, id) and the group arrays are `selectors_info`
 * (plonky2/src/gates/selectors.rs:17-26). */
typedef struct gl_circuit_desc {
    uint32_t degree_bits;              /* log2 of the trace length n                                */
    uint32_t num_wires;                /* 135                                                       */
    uint32_t num_routed_wires;         /* 80                                                        */
    uint32_t num_constants;            /* selector columns + gate-constant columns (4)              */
    uint32_t num_selectors;            /* 2                                                         */
    uint32_t num_challenges;           /* 2                                                         */
    uint32_t quotient_degree_factor;   /* 8                                                         */
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t num_fri_rounds;           /* reduction_arity_bits.len()                                */
    uint32_t fri_arity_bits[8];
    uint32_t num_public_inputs;
    uint32_t num_gates;                /* <= GL_MAX_GATES                                           */
    uint8_t gate_types[16];
    uint8_t gate_params[16];           /* LookupGate / LookupTableGate: the gate's table (index into the lists below); RandomAccessGate: its index bits; else 0 */
    uint32_t gate_selector_index[16];
    uint32_t gate_group_start[16], gate_group_end[16];
    uint64_t k_is[80];                 /* coset shifts 7^j (field/src/cosets.rs:9-24)               */
    /* ---- lookup argument (up to GL_MAX_LUTS tables; all zero without lookups).  Gate types 6 = LookupGate (40 slots), 7 =
     * LookupTableGate (26 slots) (gates/lookup.rs, gates/lookup_table.rs): one of each PER TABLE in `gate_types`, told apart by
     * `gate_params`; `num_constants` counts the lookup selector columns, which sit between the gate selectors and the gates' constants
     * (circuit_builder.rs:991-1004) ---- */
    uint32_t num_lookup_polys;         /* per challenge: 1 RE + ceil(40 / 7) partial SLDC = 7 (circuit_builder.rs:1079-1085)     */
    uint32_t num_lookup_selectors;     /* TransSre, TransLdc, InitSre, LastLdc + one end selector per table = 4 + num_luts     */
    uint32_t num_luts;                 /* common_data.luts.len(); every table has lookups (gadgets/lookup.rs:79-84)            */
    /* LookupWire per table (circuit_builder.rs:73-85; the gate rows are upside down: last_lu_row < last_lut_row <= first_lut_row).
     * ProverOnlyCircuitData: a description read from CommonCircuitData bytes has last_lu_row = 0, and build() fills the rows in
     * from the lookup selector columns (gl_circuit_description returns them). */
    uint32_t last_lu_row[4], last_lut_row[4], first_lut_row[4];
    uint32_t lut_len[4];               /* entries per table; their sum <= GL_MAX_LUT_ENTRIES                                   */
    uint16_t lut[2 * 1024];            /* (input, output) pairs (gates/lookup_table.rs:22), the tables one after the other     */
} gl_circuit_desc;
#define GL_MAX_GATES 16
#define GL_MAX_LUTS 4
#define GL_MAX_LUT_ENTRIES 1024

/* ---- context ------------------------------------------------------------------------------------ */
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or NULL to let
 * the context create its own non-blocking stream. */
int gl_ctx_create(int device, void* stream, gl_ctx** out);
/* drops the creator's reference (see "Lifetime" above); synchronises the stream first */
void gl_ctx_destroy(gl_ctx* ctx);
int gl_ctx_synchronize(gl_ctx* ctx);
/* scratch used between the two NTT passes (elements); default 2^24 (128 MiB). */
int gl_ctx_set_scratch_elems(gl_ctx* ctx, size_t elems);
const char* gl_last_error(void);
/* Per-scope device timings (HIP events on the context's stream), the analogue of the reference's TimingTree
 * (plonky2/src/util/timing.rs:8-192; scopes as in fri/oracle.rs:51-89, plonk/prover.rs:118-316).
 * Off by default; gl_ctx_timing_report synchronises and writes a JSON object into buf. */
int gl_ctx_timing_enable(gl_ctx* ctx, int on);
int gl_ctx_timing_reset(gl_ctx* ctx);
int gl_ctx_timing_report(gl_ctx* ctx, char* buf, size_t cap);
/* device memory helpers so that a host language needs no HIP binding of its own */
int gl_dev_alloc(gl_ctx* ctx, size_t bytes, void** d_out);
int gl_dev_free(gl_ctx* ctx, void* d_ptr);
int gl_copy_h2d(gl_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int gl_copy_d2h(gl_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* synchronises */

/* ---- field micro-kernels (tests; replace nothing on the path by themselves) ----------------------
 * op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 inverse(a), 5 canonicalise(a), 6 a + b*c, 7 a * 2^(b mod 192)
 * (field/src/goldilocks_field.rs:186-274,138-142).  out may alias a. */
int gl_field_op(gl_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, const uint64_t* d_c,
                uint64_t* d_out, size_t n);
/* op: 0 add, 1 sub, 2 mul, 3 inverse on interleaved (a0,a1) pairs (extension/quadratic.rs:143-193) */
int gl_ext_op(gl_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);

/* ---- NTT family: d_data is [batch][2^log_n], transformed in place, natural order in and out ------ */
/* fft_with_options(poly, None, root_table) (field/src/fft.rs:56-65): values[i] = P(w^i) */
int gl_ntt_forward(gl_ctx* ctx, uint64_t* d_data, uint32_t log_n, uint32_t batch);
/* ifft_with_options (field/src/fft.rs:72-95) */
int gl_ntt_inverse(gl_ctx* ctx, uint64_t* d_data, uint32_t log_n, uint32_t batch);
/* PolynomialCoeffs::coset_fft (field/src/polynomial/mod.rs:276-295): evaluations on shift*H */
int gl_ntt_coset_forward(gl_ctx* ctx, uint64_t* d_data, uint32_t log_n, uint32_t batch, uint64_t shift);
/* PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:58-70) */
int gl_ntt_coset_inverse(gl_ctx* ctx, uint64_t* d_data, uint32_t log_n, uint32_t batch, uint64_t shift);
/* p.lde(rate_bits).coset_fft_with_options(F::coset_shift(), Some(rate_bits), table)
 * (plonky2/src/fri/oracle.rs:111-118): d_coeffs [batch][2^log_n] -> d_out [batch][2^(log_n+rate_bits)],
 * natural order (index i <-> point 7*w^i). */
int gl_ntt_coset_lde(gl_ctx* ctx, const uint64_t* d_coeffs, uint32_t log_n, uint32_t rate_bits,
                     uint32_t batch, uint64_t* d_out);
/* host-buffer convenience (H2D + transform + D2H): direction 0 forward, 1 inverse */
int gl_fft_host(gl_ctx* ctx, uint64_t* h_data, uint32_t log_n, uint32_t batch, int inverse);

/* ---- Poseidon / hashing --------------------------------------------------------------------------*/
/* Poseidon::poseidon (plonky2/src/hash/poseidon.rs:598-609) on `count` 12-word states, in place */
int gl_poseidon_permute(gl_ctx* ctx, uint64_t* d_states, size_t count);
/* Hasher::hash_or_noop (plonky2/src/plonk/config.rs:55-66) of `count` rows of `len` elements,
 * row-major d_rows[count][len] -> d_out[count][4] */
int gl_hash_rows(gl_ctx* ctx, const uint64_t* d_rows, size_t count, size_t len, uint64_t* d_out);

/* ---- Merkle tree ---------------------------------------------------------------------------------*/
/* MerkleTree::new(leaves, cap_height) (plonky2/src/hash/merkle_tree.rs:135-165) for row-major host
 * leaves h_leaves[num_leaves][leaf_len]; the tree keeps a device copy of the leaves. */
int gl_merkle_new(gl_ctx* ctx, const uint64_t* h_leaves, size_t num_leaves, size_t leaf_len,
                  uint32_t cap_height, gl_merkle** out);
/* field `cap` (merkle_tree.rs:54): h_out[2^cap_height][4] */
int gl_merkle_cap(const gl_merkle* t, uint64_t* h_out);
/* MerkleTree::prove (merkle_tree.rs:171-207): siblings bottom-up, h_out[log2(n) - cap_height][4] */
int gl_merkle_prove(const gl_merkle* t, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings);
void gl_merkle_free(gl_merkle* t);

/* ---- PolynomialBatch -----------------------------------------------------------------------------*/
/* PolynomialBatch::from_values(values, rate_bits, blinding, cap_height, timing, fft_root_table)
 * (plonky2/src/fri/oracle.rs:43-66).  h_cols[c] points at the n = 2^k values of polynomial c.
 * blinding must be 0 (GL_ERR_UNSUPPORTED otherwise).  Keeps coefficients, LDE values and all Merkle
 * digests device-resident. */
int gl_batch_from_values(gl_ctx* ctx, const uint64_t* const* h_cols, size_t ncols, size_t n,
                         uint32_t rate_bits, uint32_t blinding, uint32_t cap_height, gl_batch** out);
/* PolynomialBatch::from_coeffs (plonky2/src/fri/oracle.rs:68-98) */
int gl_batch_from_coeffs(gl_ctx* ctx, const uint64_t* const* h_cols, size_t ncols, size_t n,
                         uint32_t rate_bits, uint32_t blinding, uint32_t cap_height, gl_batch** out);
/* same, from a device-resident column-major matrix d_cols[ncols][n]; is_values selects from_values */
int gl_batch_from_device(gl_ctx* ctx, const uint64_t* d_cols, size_t ncols, size_t n, uint32_t rate_bits,
                         uint32_t cap_height, int is_values, gl_batch** out);
/* field `merkle_tree.cap` (used at prover.rs:164,225,273,319-321): h_out[2^cap_height][4] */
int gl_batch_cap(const gl_batch* b, uint64_t* h_out);
/* merkle_tree.get(i) (merkle_tree.rs:167-169): the leaf at Merkle index i, h_out[ncols] */
int gl_batch_get_leaf(const gl_batch* b, size_t leaf_index, uint64_t* h_out);
/* PolynomialBatch::get_lde_values(index, step) (oracle.rs:128-133), h_out[ncols] */
int gl_batch_get_lde_values(const gl_batch* b, size_t index, size_t step, uint64_t* h_out);
/* merkle_tree.prove(i) (merkle_tree.rs:171-207) */
int gl_batch_prove(const gl_batch* b, size_t leaf_index, uint64_t* h_out, uint32_t* n_siblings);
/* field `polynomials` (oracle.rs:32): h_out[ncols][n] coefficients */
int gl_batch_coeffs(const gl_batch* b, uint64_t* h_out);
/* all LDE values in natural order, column-major h_out[ncols][n << rate_bits] (index i <-> 7*w^i) */
int gl_batch_lde(const gl_batch* b, uint64_t* h_out);
size_t gl_batch_ncols(const gl_batch* b);
size_t gl_batch_degree(const gl_batch* b);
const uint64_t* gl_batch_dev_coeffs(const gl_batch* b);   /* d [ncols][n]  */
const uint64_t* gl_batch_dev_lde(const gl_batch* b);      /* d [ncols][N]  */
void gl_batch_free(gl_batch* b);

/* ---- circuit data ----------------------------------------------------------------------------------*/
/* Host side of the demo (no GPU needed): the matmul circuit of plonky2/src/bin/matrix_mul.rs:25-67 after
 * CircuitBuilder::build() (plonk/circuit_builder.rs:913-1146): descriptor, gate type per row, and the
 * constants || sigmas VALUE columns, h_out[(num_constants + 80)][n]. */
int gl_matmul_circuit_build(size_t m, gl_host_circuit** out);
int gl_host_circuit_desc(const gl_host_circuit* hc, gl_circuit_desc* out);
int gl_host_circuit_row_gates(const gl_host_circuit* hc, uint8_t* h_out /* n */);
int gl_host_circuit_constants_sigmas(const gl_host_circuit* hc, uint64_t* h_out);
/* generate_partial_witness + full_witness (plonk/prover.rs:118-133) for this circuit: a, b row-major m x m;
 * the 131 values the reference draws from OsRng for the unused PublicInputGate wires
 * (circuit_builder.rs:904-910) come from splitmix64(filler_seed).  h_wires[135][n], h_public_inputs[3 m^2]. */
int gl_matmul_witness(const gl_host_circuit* hc, const uint64_t* h_a, const uint64_t* h_b, uint64_t filler_seed,
                      uint64_t* h_wires, uint64_t* h_public_inputs);
void gl_host_circuit_free(gl_host_circuit* hc);
/* The same witness produced directly in HBM: the m^3 ArithmeticGate operations are filled by the GPU, the sequential
 * public-input hash sponge (PoseidonGate rows) by the calling host thread meanwhile.  d_wires[135][n] is overwritten;
 * h_public_inputs[3 m^2].  The generator borrows `hc` and belongs to `ctx` (one per context / stream). */
typedef struct gl_matmul_witgen gl_matmul_witgen;
int gl_matmul_witgen_create(gl_ctx* ctx, const gl_host_circuit* hc, gl_matmul_witgen** out);
int gl_matmul_witgen_run(gl_matmul_witgen* g, const uint64_t* a, const uint64_t* b, uint64_t filler_seed,
                         uint64_t* d_wires, uint64_t* h_public_inputs, uint64_t* h_public_inputs_hash /* [4], may be null */);
void gl_matmul_witgen_free(gl_matmul_witgen* g);

/* The device half of build(): PolynomialBatch::from_values(constants || sigmas) (circuit_builder.rs:1020-1028),
 * circuit_digest (:1089-1100), sigma / subgroup tables.  h_constants_sigmas[(num_constants + 80)][n]. */
int gl_circuit_create(gl_ctx* ctx, const gl_circuit_desc* desc, const uint64_t* h_constants_sigmas, gl_circuit** out);
/* The same with the sigma polynomials computed ON THE DEVICE (circuit_builder.rs:1007-1014 sigma_vecs; plonk/permutation_argument.rs:
 * 85-170): h_wire_classes[80][n] holds, for every routed wire (wire (row, col) at col * n + row), the id of its copy-constraint
 * class -- the representative the reference's union-find Forest assigns; any u64, equal ids = wires constrained equal.
 * h_constants[num_constants][n] are the selector and gate-constant VALUE columns. */
int gl_circuit_create_from_classes(gl_ctx* ctx, const gl_circuit_desc* desc, const uint64_t* h_constants, const uint64_t* h_wire_classes,
                                   gl_circuit** out);
/* build() of the demo circuit: constants from the host description, sigma polynomials on the device */
int gl_circuit_from_host(gl_ctx* ctx, const gl_host_circuit* hc, gl_circuit** out);
/* the wire classes of the host description, h_out[80][n] (gl_circuit_create_from_classes takes them) */
int gl_host_circuit_wire_classes(const gl_host_circuit* hc, uint64_t* h_out);
/* The description the circuit was built with, completed: for a circuit with lookups build() reads last_lu_row / last_lut_row /
 * first_lut_row from the lookup selector columns (they are ProverOnlyCircuitData::lookup_rows in the reference and are not part of
 * CommonCircuitData's bytes), checks any the caller named against them, and refuses a mismatch with GL_ERR_ARG. */
int gl_circuit_description(const gl_circuit* c, gl_circuit_desc* out);
/* Optional, once per (context, circuit) before the first proof: one pass of the proving pipeline over a zero witness, thrown away,
 * so that the first gl_prove on this context does not pay for loading the kernels' code objects, building the twiddle tables of the
 * circuit's transform sizes and growing the context's pool (m = 64: 15 ms for the first proof without it, 7 ms with). */
int gl_circuit_warm_up(gl_ctx* ctx, const gl_circuit* c);
int gl_circuit_digest(const gl_circuit* c, uint64_t h_out[4]);                 /* verifier_only.circuit_digest */
int gl_circuit_constants_sigmas_cap(const gl_circuit* c, uint64_t* h_out);     /* [2^cap_height][4]            */
const gl_batch* gl_circuit_constants_sigmas_batch(const gl_circuit* c);
void gl_circuit_free(gl_circuit* c);

/* ---- prover phases ---------------------------------------------------------------------------------*/
/* The seam for a caller that keeps the Fiat-Shamir transcript (iop/challenger.rs) on its own side: each entry
 * point replaces one call of plonk::prover::prove and returns exactly what the transcript absorbs next.
 * gl_prove() below is these calls in the order of prover.rs:102-329 with the Challenger in C++.
 * All polynomial-sized data stays in HBM; challenges may be non-canonical u64, outputs are canonical. */
typedef struct gl_fri gl_fri;

/* all_wires_permutation_partial_products (plonk/prover.rs:189-200,332-416) followed by the commitment of
 * prover.rs:212-223: d_wires[135][n] witness VALUES on the device -> PolynomialBatch of the 2 Z and 18 partial
 * product polynomials (column order zs, then partial products, as prover.rs:202-210). */
int gl_partial_products(gl_ctx* ctx, const gl_circuit* c, const uint64_t* d_wires, const uint64_t betas[2],
                        const uint64_t gammas[2], gl_batch** out);
/* compute_quotient_polys + the split into quotient_degree_factor chunks + from_coeffs (plonk/prover.rs:229-271,
 * 574-744; vanishing_poly.rs:164-330): PolynomialBatch of the 2 x 8 chunk polynomials. */
int gl_quotient_polys(gl_ctx* ctx, const gl_circuit* c, const gl_batch* wires, const gl_batch* zs_partial_products,
                      const uint64_t public_inputs_hash[4], const uint64_t betas[2], const uint64_t gammas[2],
                      const uint64_t alphas[2], gl_batch** out);
/* The same two phases for a circuit WITH the lookup argument: `deltas[8]` = per challenge ChallengeA, ChallengeB, ChallengeAlpha,
 * ChallengeDelta, i.e. [betas | gammas | the 4 challenges drawn after them] (plonk/prover.rs:166-184).  The batch of the first has the
 * 2 x 7 lookup polynomials behind the 20 columns (compute_all_lookup_polys, prover.rs:202-211); the second adds
 * check_lookup_constraints (vanishing_poly.rs:503-670) to the quotient.  gl_fri_combine and gl_open_at need no variant: the batch
 * carries its columns (openings in FriOpenings order: ..., quotient, lookups | zs_next, lookups_next, plonk/proof.rs:346-380). */
int gl_partial_products_lookups(gl_ctx* ctx, const gl_circuit* c, const uint64_t* d_wires, const uint64_t betas[2],
                                const uint64_t gammas[2], const uint64_t deltas[8], gl_batch** out);
int gl_quotient_polys_lookups(gl_ctx* ctx, const gl_circuit* c, const gl_batch* wires, const gl_batch* zs_partial_products_lookups,
                              const uint64_t public_inputs_hash[4], const uint64_t betas[2], const uint64_t gammas[2],
                              const uint64_t alphas[2], const uint64_t deltas[8], gl_batch** out);
/* OpeningSet::new's eval_commitment (plonk/proof.rs:306-344): polynomials first_col .. first_col + num_cols of a
 * batch at the extension point z; h_out[num_cols][2]. */
int gl_open_at(gl_ctx* ctx, const gl_batch* b, const uint64_t z[2], size_t first_col, size_t num_cols, uint64_t* h_out);
/* PolynomialBatch::prove_openings up to the call of fri_proof (fri/oracle.rs:162-204): batches in oracle order
 * constants||sigmas, wires, Z||partial products, quotient (circuit_data.rs:586-595); opens everything at zeta and
 * the Z polynomials at g*zeta, final = alpha^2 Q0 + Q1, LDE onto the coset.  The batches must outlive the gl_fri. */
int gl_fri_combine(gl_ctx* ctx, const gl_circuit* c, const gl_batch* const batches[4], const uint64_t zeta[2],
                   const uint64_t alpha[2], gl_fri** out);
/* fri_committed_trees, one loop iteration in two halves (fri/prover.rs:76-103): Merkle tree of the current
 * codeword -> its cap h_cap_out[2^cap_height][4]; then, with the beta drawn after observing that cap, the fold. */
int gl_fri_commit_round(gl_fri* f, uint64_t* h_cap_out);
int gl_fri_fold(gl_fri* f, const uint64_t beta[2]);
/* final_poly (fri/prover.rs:106-111): *num_words = 2 * len; h_out (may be null to query the size) = (a, b) pairs */
int gl_fri_final_poly(gl_fri* f, uint64_t* h_out, size_t cap_words, size_t* num_words);
/* fri_proof_of_work (fri/prover.rs:115-160): the SMALLEST witness w with
 * leading_zeros(permute(sponge_state overlaid with input_buffer[0..input_len) and w at input_len)[7]) >= min_leading_zeros */
int gl_pow_grind(gl_ctx* ctx, const uint64_t sponge_state[12], const uint64_t* input_buffer, uint32_t input_len,
                 uint32_t min_leading_zeros, uint64_t* witness);
/* fri_prover_query_rounds (fri/prover.rs:162-216) for the given x_index values (challenge mod lde_size), as the
 * serialised Vec<FriQueryRound> body (util/serialization/mod.rs:1477-1546).  h_blob may be null to query the size. */
int gl_fri_query(gl_fri* f, const uint32_t* x_index, uint32_t num_queries, uint8_t* h_blob, size_t cap_bytes,
                 size_t* num_bytes);
void gl_fri_free(gl_fri* f);

/* A Challenger (iop/challenger.rs:30-153: duplex sponge, challenges pop from the end of the rate) for callers of the
 * phase API that have no transcript of their own.  Host code.  `gl_challenger_state` exposes sponge_state / input_buffer
 * the way fri_proof_of_work reads them (fri/prover.rs:127-140), for gl_pow_grind. */
typedef struct gl_challenger gl_challenger;
gl_challenger* gl_challenger_new(void);
int gl_challenger_observe(gl_challenger* c, const uint64_t* h_elements, size_t count);
int gl_challenger_get_challenges(gl_challenger* c, uint64_t* h_out, size_t count);
int gl_challenger_state(const gl_challenger* c, uint64_t h_sponge_state[12], uint64_t h_input_buffer[8], uint32_t* input_len);
void gl_challenger_free(gl_challenger* c);

/* ---- prove() ---------------------------------------------------------------------------------------*/
/* plonk::prover::prove (plonky2/src/plonk/prover.rs:102-329) from step 4 on, i.e. given the FULL witness matrix
 * `MatrixWitness.wire_values` (iop/witness.rs:256-258) h_wires[num_wires][n] and the public inputs.  Every
 * polynomial stays device-resident; the host only sees Merkle caps, openings and query answers.  The PoW
 * witness is the smallest valid one (the 1-thread order of fri/prover.rs:141-152).
 * Returns GL_ERR_ZETA_IN_SUBGROUP for prover.rs:280-283. */
int gl_prove(gl_ctx* ctx, const gl_circuit* c, const uint64_t* h_wires, const uint64_t* h_public_inputs,
             size_t num_public_inputs, gl_proof** out);
/* same with the witness as the reference stores it, one host vector per wire: h_wire_columns[num_wires] -> n values each
 * (`MatrixWitness.wire_values: Vec<Vec<F>>`), so that the caller does not have to flatten 135 vectors first */
int gl_prove_columns(gl_ctx* ctx, const gl_circuit* c, const uint64_t* const* h_wire_columns,
                     const uint64_t* h_public_inputs, size_t num_public_inputs, gl_proof** out);
/* same with the witness matrix already resident in HBM (d_wires[num_wires][n]) */
int gl_prove_device(gl_ctx* ctx, const gl_circuit* c, const uint64_t* d_wires, const uint64_t* h_public_inputs,
                    size_t num_public_inputs, gl_proof** out);
/* same, with public_inputs_hash = hash_no_pad(public_inputs) (prover.rs:126-127) supplied by the caller -- the witness
 * generator's sponge produces it as a by-product, and hashing 3 m^2 inputs is a sequential host job */
int gl_prove_device_hashed(gl_ctx* ctx, const gl_circuit* c, const uint64_t* d_wires, const uint64_t* h_public_inputs,
                           size_t num_public_inputs, const uint64_t public_inputs_hash[4], gl_proof** out);
/* Many independent proofs in flight on one GPU from one call -- what the reference gets from its Rayon pool when a batch of
 * witnesses is proved.  The pool owns one device-resident circuit and `lanes` contexts (HIP stream, allocator, witness
 * generator each); item i is proved on lane i % lanes by `lanes` host threads inside the call.  `hc` is borrowed. */
typedef struct gl_prover_pool gl_prover_pool;
int gl_prover_pool_create(int device, const gl_host_circuit* hc, uint32_t lanes, gl_prover_pool** out);
uint32_t gl_prover_pool_lanes(const gl_prover_pool* p);
const gl_circuit* gl_prover_pool_circuit(const gl_prover_pool* p);     /* for gl_circuit_digest / _constants_sigmas_cap */
/* a[i], b[i]: row-major m x m operands on the host; filler_seeds may be null (seed i); out_proofs[count] */
int gl_prover_pool_prove_matmul(gl_prover_pool* p, size_t count, const uint64_t* const* a, const uint64_t* const* b,
                                const uint64_t* filler_seeds, gl_proof** out_proofs);
/* The pool for ANY circuit the library proves (what gl_circuit_create takes), every lane warmed up (gl_circuit_warm_up), and `count`
 * proofs from HOST witnesses on it: columns[i][j] = wire column j (n values) of witness i, as MatrixWitness.wire_values holds them
 * (iop/witness.rs:256-258), public_inputs[i] = its num_public_inputs values; item i goes through gl_prove_columns on lane i % lanes.
 * The batch analogue of the INTEGRATION.md section-3 patch: a Rayon `par_iter` over witnesses becomes one call. */
int gl_prover_pool_create_generic(int device, const gl_circuit_desc* desc, const uint64_t* h_constants_sigmas, uint32_t lanes, gl_prover_pool** out);
int gl_prover_pool_prove_columns(gl_prover_pool* p, size_t count, const uint64_t* const* const* columns, const uint64_t* const* public_inputs,
                                 gl_proof** out_proofs);
void gl_prover_pool_free(gl_prover_pool* p);

/* ProofWithPublicInputs::to_bytes (plonk/proof.rs:104-110; util/serialization/mod.rs:1939-1981) */
size_t gl_proof_num_bytes(const gl_proof* p);
int gl_proof_bytes(const gl_proof* p, uint8_t* h_out, size_t cap);
/* intermediates, for parity tests: betas[2] gammas[2] alphas[2] zeta[2] fri_alpha[2] pow_witness pi_hash[4],
 * then the FRI betas (2 words each); returns the number of words written */
size_t gl_proof_challenges(const gl_proof* p, uint64_t* h_out);
int gl_proof_caps(const gl_proof* p, uint64_t* h_out /* [3][2^cap_height][4]: wires, zs_pp, quotient */);
/* the next two need gl_ctx_capture_intermediates(ctx, 1) before proving (two extra device->host copies per proof) */
int gl_ctx_capture_intermediates(gl_ctx* ctx, int enable);
int gl_proof_zs_partial_products(const gl_proof* p, uint64_t* h_out /* [20][n] values; [34][n] with lookups: the 14 lookup polynomials follow */);
int gl_proof_quotient_chunks(const gl_proof* p, uint64_t* h_out /* [16][n] coefficients */);
size_t gl_proof_query_indices(const gl_proof* p, uint64_t* h_out);
void gl_proof_free(gl_proof* p);

/* ---- verify() --------------------------------------------------------------------------------------*/
/* VerifierCircuitData::verify (plonky2/src/plonk/circuit_data.rs:208-215 -> plonk/verifier.rs:15-115, fri/verifier.rs:
 * 62-260) for circuits over the demo's gate set: CommonCircuitData = *desc, VerifierOnlyCircuitData =
 * (constants_sigmas_cap[2^cap_height][4], circuit_digest), proof = ProofWithPublicInputs::to_bytes().  Host code, no
 * GPU needed (as in the reference).  GL_OK = accepted; GL_ERR_VERIFY = rejected, gl_last_error() names the failing
 * check ("vanishing polynomial identity fails at zeta", "invalid proof of work witness", "initial Merkle proof fails",
 * "FRI consistency check fails", "FRI step Merkle proof fails", "final polynomial evaluation is invalid",
 * "malformed proof: ..."). */
int gl_verify(const gl_circuit_desc* desc, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4],
              const uint8_t* proof_bytes, size_t num_bytes);
int gl_host_circuit_verify(const gl_host_circuit* hc, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4],
                           const uint8_t* proof_bytes, size_t num_bytes);

/* ---- circuit data as bytes --------------------------------------------------------------------------*/
/* The reference's serialised forms (plonky2/src/plonk/circuit_data.rs:125-142,208-238; util/serialization/mod.rs:739-800,
 * 1736-1790 CommonCircuitData, :909-930,1889-1906 VerifierOnlyCircuitData, :1908-1919 VerifierCircuitData = verifier_only ||
 * common; gates tagged as by DefaultGateSerializer, util/serialization/gate_serialization.rs:87-108), so that a Rust-built
 * circuit's CommonCircuitData / VerifierCircuitData bytes can be handed over instead of a hand-filled gl_circuit_desc.  Host code.
 * Representable: the demo's five gates, standard_recursion_config's shape (no lookups, no zero-knowledge); anything else
 * is GL_ERR_UNSUPPORTED when reading.  h_out may be null to query *num_bytes.  The value columns (constants, sigmas) and the
 * generators of ProverOnlyCircuitData are not part of these forms: gl_circuit_create still takes the columns. */
int gl_common_data_to_bytes(const gl_circuit_desc* desc, uint8_t* h_out, size_t cap, size_t* num_bytes);
int gl_common_data_from_bytes(const uint8_t* h_bytes, size_t num_bytes, gl_circuit_desc* out, size_t* consumed /* may be null */);
int gl_verifier_only_to_bytes(uint32_t cap_height, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4], uint8_t* h_out,
                              size_t cap, size_t* num_bytes);
int gl_verifier_only_from_bytes(const uint8_t* h_bytes, size_t num_bytes, uint32_t* cap_height, uint64_t* h_cap /* [2^h][4], may be null */,
                                size_t cap_words, uint64_t circuit_digest[4], size_t* consumed /* may be null */);
/* VerifierCircuitData::from_bytes(data).verify(proof): GL_OK = accepted, GL_ERR_VERIFY = rejected (as gl_verify) */
int gl_verify_bytes(const uint8_t* h_verifier_data, size_t num_data_bytes, const uint8_t* proof_bytes, size_t num_proof_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PLONKY2_MI355X_H */
