// prove(): host driver replicating the phase order and Fiat-Shamir transcript of plonky2/src/plonk/prover.rs:102-329
// with every polynomial-sized object resident in HBM.  The host sees only Merkle caps, openings, the final FRI
// polynomial, the PoW witness and the query answers; each of those is a mandatory sync point because the next
// phase's challenge is a Poseidon transcript of it (iop/challenger.rs:81-148).
#include "context.hpp"
#include "host_circuit.hpp"
#include "prover_kernels.cuh"
#include <cstring>
#include <memory>

using glhost::HostCircuit;

struct gl_circuit {
    gl_ctx* ctx = nullptr;
    gl_circuit_desc desc;
    size_t n = 0;
    gl_batch* cs_batch = nullptr;     // constants || sigmas commitment
    gl_t* d_sigmas = nullptr;         // sigma VALUES [80][n]
    gl_t* d_l0_coset = nullptr;       // L_0 on the LDE coset [8n]
    gl_t circuit_digest[4];
};

struct gl_proof {
    std::vector<uint8_t> bytes;
    std::vector<gl_t> challenges;     // betas gammas alphas zeta fri_alpha pow pi_hash fri_betas...
    std::vector<gl_t> caps;           // 3 x 16 x 4
    std::vector<gl_t> zs_pp;          // [20][n]
    std::vector<gl_t> quotient;       // [16][n]
    std::vector<uint64_t> query_indices;
};

// ---- host Challenger (iop/challenger.rs:30-153) -----------------------------------------------------------------
struct HostChallenger {
    gl_t state[12]; gl_t in[8]; int nin = 0; gl_t out[8]; int nout = 0;
    HostChallenger() { for (auto& s : state) s = 0; }
    void duplexing() {
        for (int i = 0; i < nin; i++) state[i] = in[i];
        nin = 0;
        psd_permute(state);
        for (int i = 0; i < 8; i++) out[i] = state[i];
        nout = 8;
    }
    void observe(gl_t x) { nout = 0; in[nin++] = x; if (nin == 8) duplexing(); }
    void observe_many(const gl_t* v, size_t n) { for (size_t i = 0; i < n; i++) observe(v[i]); }
    gl_t challenge() { if (nin || !nout) duplexing(); return gl_canon(out[--nout]); }
};

// C handle of the Challenger for callers of the phase API that have no transcript of their own (C / C++ / Python)
struct gl_challenger { HostChallenger ch; };
extern "C" gl_challenger* gl_challenger_new(void) { return new gl_challenger(); }
extern "C" void gl_challenger_free(gl_challenger* c) { delete c; }
extern "C" int gl_challenger_observe(gl_challenger* c, const uint64_t* h_elements, size_t count) {
    GL_REQUIRE(c && (h_elements || !count), GL_ERR_ARG, "gl_challenger_observe: null argument");
    c->ch.observe_many(h_elements, count);
    return GL_OK;
}
extern "C" int gl_challenger_get_challenges(gl_challenger* c, uint64_t* h_out, size_t count) {
    GL_REQUIRE(c && (h_out || !count), GL_ERR_ARG, "gl_challenger_get_challenges: null argument");
    for (size_t i = 0; i < count; i++) h_out[i] = c->ch.challenge();
    return GL_OK;
}
// sponge state and pending inputs, as fri_proof_of_work reads them (fri/prover.rs:127-140): for gl_pow_grind
extern "C" int gl_challenger_state(const gl_challenger* c, uint64_t h_sponge_state[12], uint64_t h_input_buffer[8], uint32_t* input_len) {
    GL_REQUIRE(c && h_sponge_state && h_input_buffer && input_len, GL_ERR_ARG, "gl_challenger_state: null argument");
    for (int i = 0; i < 12; i++) h_sponge_state[i] = c->ch.state[i];
    for (int i = 0; i < c->ch.nin; i++) h_input_buffer[i] = c->ch.in[i];
    *input_len = (uint32_t)c->ch.nin;
    return GL_OK;
}

// ---- host circuit API ------------------------------------------------------------------------------------------------
extern "C" int gl_matmul_circuit_build(size_t m, gl_host_circuit** out) {
    GL_REQUIRE(out, GL_ERR_ARG, "null out");
    std::unique_ptr<gl_host_circuit> h(new gl_host_circuit());
    int st = glhost::build_matmul(m, &h->hc);
    if (st != GL_OK) return gl_fail(st, "matmul dimension out of range (1..256)", __FILE__, __LINE__);
    *out = h.release();
    return GL_OK;
}
extern "C" int gl_host_circuit_desc(const gl_host_circuit* hc, gl_circuit_desc* out) {
    GL_REQUIRE(hc && out, GL_ERR_ARG, "null argument");
    *out = hc->hc.desc;
    return GL_OK;
}
extern "C" int gl_host_circuit_row_gates(const gl_host_circuit* hc, uint8_t* h_out) {
    GL_REQUIRE(hc && h_out, GL_ERR_ARG, "null argument");
    memcpy(h_out, hc->hc.row_gate.data(), hc->hc.row_gate.size());
    return GL_OK;
}
extern "C" int gl_host_circuit_constants_sigmas(const gl_host_circuit* hc, uint64_t* h_out) {
    GL_REQUIRE(hc && h_out, GL_ERR_ARG, "null argument");
    hc->hc.ensure_host_sigmas();
    memcpy(h_out, hc->hc.constants_sigmas.data(), hc->hc.constants_sigmas.size() * sizeof(gl_t));
    return GL_OK;
}
extern "C" int gl_matmul_witness(const gl_host_circuit* hc, const uint64_t* a, const uint64_t* b, uint64_t filler_seed, uint64_t* h_wires, uint64_t* h_pis) {
    GL_REQUIRE(hc && a && b && h_wires && h_pis, GL_ERR_ARG, "null argument");
    return glhost::matmul_witness(hc->hc, a, b, filler_seed, h_wires, h_pis);
}
extern "C" void gl_host_circuit_free(gl_host_circuit* hc) { delete hc; }

// ---- device circuit -------------------------------------------------------------------------------------------------
static int validate_desc(const gl_circuit_desc& d) {
    GL_REQUIRE(d.num_wires == 135 && d.num_routed_wires == 80 && d.num_challenges == 2 && d.quotient_degree_factor == 8, GL_ERR_UNSUPPORTED,
               "only standard_recursion_config (135 wires, 80 routed, 2 challenges, quotient factor 8) is supported");
    GL_REQUIRE(d.rate_bits == 3, GL_ERR_UNSUPPORTED, "rate_bits must equal log2(quotient_degree_factor) = 3 (prover.rs:596-608 step = 1)");
    GL_REQUIRE(d.cap_height <= d.degree_bits + d.rate_bits && d.degree_bits >= 1 && d.degree_bits + d.rate_bits <= 24, GL_ERR_ARG, "bad degree / cap height");
    GL_REQUIRE(d.num_gates >= 1 && d.num_gates <= GL_MAX_GATES && d.num_selectors >= 1 && d.num_selectors <= 4, GL_ERR_ARG, "bad gate / selector count");
    { const char* why = glhost::lookup_shape_error(d); GL_REQUIRE(!why, GL_ERR_UNSUPPORTED, why); }
    const uint32_t nrows = 1u << d.degree_bits;
    for (unsigned t = 0; t < d.num_luts; t++)
        GL_REQUIRE(d.last_lu_row[t] < d.last_lut_row[t] && d.last_lut_row[t] <= d.first_lut_row[t] && d.first_lut_row[t] + 1 < nrows &&
                   d.first_lut_row[t] - d.last_lut_row[t] + 1 == glhost::lut_rows(d, t) && (t == 0 || d.last_lu_row[t] > d.first_lut_row[t - 1] + 1),
                   GL_ERR_ARG, "bad lookup rows");
    GL_REQUIRE(d.num_constants == d.num_selectors + d.num_lookup_selectors + 2, GL_ERR_ARG, "bad gate / selector description");
    GL_REQUIRE(d.num_fri_rounds <= 8 && d.num_query_rounds >= 1 && d.num_query_rounds <= 256 && d.proof_of_work_bits <= 40, GL_ERR_ARG, "bad FRI parameters");
    unsigned tot = 0;
    for (unsigned r = 0; r < d.num_fri_rounds; r++) { GL_REQUIRE(d.fri_arity_bits[r] == 4, GL_ERR_UNSUPPORTED, "FRI arity must be 16"); tot += 4; }
    GL_REQUIRE(tot <= d.degree_bits && d.degree_bits + d.rate_bits >= tot + d.cap_height, GL_ERR_ARG, "FRI total reduction arity is too large");   // circuit_builder.rs:977-980
    for (unsigned g = 0; g < d.num_gates; g++) {
        GL_REQUIRE(d.gate_types[g] <= glhost::G_LAST, GL_ERR_UNSUPPORTED, "gate type not in {Noop, Constant, PublicInput, Arithmetic, Poseidon, BaseSum<2>, Lookup, LookupTable, Exponentiation, RandomAccess}");
        GL_REQUIRE(d.gate_selector_index[g] < d.num_selectors && d.gate_group_start[g] <= g && g < d.gate_group_end[g] && d.gate_group_end[g] <= d.num_gates, GL_ERR_ARG, "bad selector group");
    }
    return GL_OK;
}

extern "C" void gl_circuit_free(gl_circuit* c) {
    if (!c) return;
    if (c->cs_batch) gl_batch_free(c->cs_batch);
    if (c->d_sigmas) c->ctx->pool_release(c->d_sigmas);
    if (c->d_l0_coset) c->ctx->pool_release(c->d_l0_coset);
    gl_ctx_release(c->ctx);
    delete c;
}

// The lookup rows of a description are ProverOnlyCircuitData (lookup_rows, circuit_data.rs:335-360): CommonCircuitData's bytes do not carry
// last_lu_row.  The lookup SELECTOR columns do (gates/selectors.rs:50-103): table t's end selector is 1 at last_lut_row[t] only; LastLdc is 1
// at every table's last_lu_row and InitSre at every first_lut_row + 1, and a table's rows lie between the previous table's and the next
// one's.  build() reads the rows from there and refuses a description that disagrees.
static int lookup_rows_from_selectors(gl_circuit_desc& d, const uint64_t* h_constants) {
    if (!d.num_luts) return GL_OK;
    { const char* why = glhost::lookup_shape_error(d); GL_REQUIRE(!why, GL_ERR_UNSUPPORTED, why); }
    GL_REQUIRE(d.num_constants >= d.num_selectors + d.num_lookup_selectors, GL_ERR_ARG, "bad gate / selector description");
    const size_t n = size_t(1) << d.degree_bits;
    auto ones = [&](unsigned sel, std::vector<uint32_t>& rows) {            // the rows where a 0/1 selector column is 1, ascending
        const uint64_t* col = h_constants + (size_t)(d.num_selectors + sel) * n;
        for (size_t r = 0; r < n; r++) if (col[r] == 1) rows.push_back((uint32_t)r); else if (col[r] != 0) return false;
        return true;
    };
    std::vector<uint32_t> last_ldc, init_sre;
    GL_REQUIRE(ones(glhost::LU_SEL_LAST_LDC, last_ldc) && ones(glhost::LU_SEL_INIT_SRE, init_sre) && last_ldc.size() == d.num_luts && init_sre.size() == d.num_luts,
               GL_ERR_ARG, "lookup selector columns do not describe num_luts tables");
    for (unsigned t = 0; t < d.num_luts; t++) {
        std::vector<uint32_t> end;
        GL_REQUIRE(ones(glhost::LU_SEL_START_END + t, end) && end.size() == 1, GL_ERR_ARG, "a table's end selector must be 1 on exactly one row");
        // tables are placed in order (gadgets/lookup.rs:79-125): the t-th LastLdc / InitSre rows are table t's
        const uint32_t lu = last_ldc[t], lut = end[0], first = init_sre[t] - 1;
        GL_REQUIRE(init_sre[t] >= 1 && lu < lut && lut <= first, GL_ERR_ARG, "lookup selector columns do not describe the tables in order");
        // a description read from common-data bytes has last_lu_row = 0 (unknown); any row it does name has to agree
        GL_REQUIRE((!d.last_lu_row[t] || d.last_lu_row[t] == lu) && (!d.last_lut_row[t] || d.last_lut_row[t] == lut) && (!d.first_lut_row[t] || d.first_lut_row[t] == first),
                   GL_ERR_ARG, "the description's lookup rows disagree with the lookup selector columns");
        d.last_lu_row[t] = lu; d.last_lut_row[t] = lut; d.first_lut_row[t] = first;
    }
    return GL_OK;
}

int gl_sigmas_from_classes(gl_ctx* c, const uint64_t* d_classes, uint32_t lgn, uint32_t ncols, const uint64_t* h_k_is, gl_t* d_sigma);      // sigma.hip

// the rest of the device half of build() once the constants || sigmas VALUE columns are in HBM (d_cs[num_constants + 80][n])
static int circuit_finish(gl_ctx* ctx, const gl_circuit_desc* desc, const gl_t* d_cs, gl_circuit** out) {
    std::unique_ptr<gl_circuit, void (*)(gl_circuit*)> c(new gl_circuit(), gl_circuit_free);
    c->ctx = ctx; ctx->retain(); c->desc = *desc; c->n = size_t(1) << desc->degree_bits;
    const size_t n = c->n, ncs = desc->num_constants + 80;
    GL_TRY(gl_batch_from_device(ctx, d_cs, ncs, n, desc->rate_bits, desc->cap_height, 1, &c->cs_batch));      // circuit_builder.rs:1020-1028
    GL_TRY(ctx->pool_alloc(80 * n * sizeof(gl_t), (void**)&c->d_sigmas));
    GL_CHECK_HIP(hipMemcpyAsync(c->d_sigmas, d_cs + (size_t)desc->num_constants * n, 80 * n * sizeof(gl_t), hipMemcpyDeviceToDevice, ctx->stream));
    {   // L_0 on the coset 7 H_N, N = n << rate_bits
        const uint32_t lgN = desc->degree_bits + desc->rate_bits;
        const size_t N = size_t(1) << lgN;
        GL_TRY(ctx->pool_alloc(N * sizeof(gl_t), (void**)&c->d_l0_coset));
        GlPowTable xt;
        GL_TRY(ctx->get_pow_table(gl_host_root_of_unity(lgN), GL_MULT_GENERATOR, (uint32_t)((N + 2047) >> 11), &xt));
        gl_t zh[8];
        gl_t g_pow_n = GL_MULT_GENERATOR; for (uint32_t i = 0; i < desc->degree_bits; i++) g_pow_n = gl_sqr(g_pow_n);
        gl_t w8 = gl_host_root_of_unity(desc->rate_bits), x = 1;
        for (int i = 0; i < 8; i++) { zh[i] = gl_canon(gl_sub(gl_mul(g_pow_n, x), 1)); x = gl_mul(x, w8); }
        GL_TRY(ctx->ensure_dev_small(1 << 20));
        GL_TRY(gl_copy_h2d(ctx, ctx->dev_small, zh, sizeof zh));
        hipLaunchKernelGGL(k_l0_on_coset, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, xt.lo, xt.hi, (uint32_t)N, (gl_t)n, ctx->dev_small, c->d_l0_coset);
        GL_CHECK_HIP(hipGetLastError());
        GL_CHECK_HIP(gl_stream_wait(ctx->stream));
    }
    // circuit_digest = hash_no_pad(cap || hash_pad([]) || [degree_bits])   (circuit_builder.rs:1089-1100)
    std::vector<gl_t> parts((size_t(4) << desc->cap_height));
    GL_TRY(gl_batch_cap(c->cs_batch, parts.data()));
    gl_t padded[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, ds[4];                   // plonk/config.rs:41-51
    glhost::host_hash_no_pad(padded, 12, ds);
    for (int k = 0; k < 4; k++) parts.push_back(ds[k]);
    parts.push_back(desc->degree_bits);
    glhost::host_hash_no_pad(parts.data(), parts.size(), c->circuit_digest);
    *out = c.release();
    return GL_OK;
}
extern "C" int gl_circuit_create(gl_ctx* ctx, const gl_circuit_desc* desc, const uint64_t* h_cs, gl_circuit** out) {
    GL_REQUIRE(ctx && desc && h_cs && out, GL_ERR_ARG, "null argument");
    GL_REQUIRE(desc->degree_bits >= 1 && desc->degree_bits <= 21 && desc->num_selectors <= 4, GL_ERR_ARG, "bad degree / selector count");
    gl_circuit_desc d = *desc;
    GL_TRY(lookup_rows_from_selectors(d, h_cs));
    GL_TRY(validate_desc(d));
    GL_TRY(ctx->activate());
    const size_t n = size_t(1) << d.degree_bits, ncs = d.num_constants + 80;
    DevBuf d_cs(ctx); GL_TRY(d_cs.alloc(ncs * n * sizeof(gl_t)));
    GL_TRY(gl_copy_h2d(ctx, d_cs.p, h_cs, ncs * n * sizeof(gl_t)));
    return circuit_finish(ctx, &d, d_cs.as<gl_t>(), out);
}
// build() with the sigma polynomials computed on the device from the copy-constraint classes (sigma.hip)
extern "C" int gl_circuit_create_from_classes(gl_ctx* ctx, const gl_circuit_desc* desc, const uint64_t* h_constants, const uint64_t* h_wire_classes, gl_circuit** out) {
    GL_REQUIRE(ctx && desc && h_constants && h_wire_classes && out, GL_ERR_ARG, "null argument");
    GL_REQUIRE(desc->degree_bits >= 1 && desc->degree_bits <= 21 && desc->num_selectors <= 4, GL_ERR_ARG, "bad degree / selector count");
    gl_circuit_desc dd = *desc;
    GL_TRY(lookup_rows_from_selectors(dd, h_constants));
    desc = &dd;
    GL_TRY(validate_desc(*desc));
    GL_TRY(ctx->activate());
    const size_t n = size_t(1) << desc->degree_bits, nc = desc->num_constants;
    DevBuf d_cs(ctx), d_cls(ctx);
    GL_TRY(d_cs.alloc((nc + 80) * n * sizeof(gl_t)));
    GL_TRY(d_cls.alloc(80 * n * sizeof(uint64_t)));
    GL_TRY(gl_copy_h2d(ctx, d_cs.p, h_constants, nc * n * sizeof(gl_t)));
    GL_TRY(gl_copy_h2d(ctx, d_cls.p, h_wire_classes, 80 * n * sizeof(uint64_t)));
    ctx->timing_begin("sigma polynomials");
    int st = gl_sigmas_from_classes(ctx, d_cls.as<uint64_t>(), desc->degree_bits, 80, desc->k_is, d_cs.as<gl_t>() + nc * n);
    ctx->timing_end();
    GL_TRY(st);
    return circuit_finish(ctx, desc, d_cs.as<gl_t>(), out);
}
extern "C" int gl_circuit_from_host(gl_ctx* ctx, const gl_host_circuit* hc, gl_circuit** out) {
    GL_REQUIRE(hc, GL_ERR_ARG, "null host circuit");
    // the constant columns are the first num_constants columns of the host matrix; the sigma columns come from the classes
    return gl_circuit_create_from_classes(ctx, &hc->hc.desc, hc->hc.constants_sigmas.data(), hc->hc.wire_class.data(), out);
}
extern "C" int gl_host_circuit_wire_classes(const gl_host_circuit* hc, uint64_t* h_out) {
    GL_REQUIRE(hc && h_out, GL_ERR_ARG, "null argument");
    memcpy(h_out, hc->hc.wire_class.data(), hc->hc.wire_class.size() * sizeof(uint64_t));
    return GL_OK;
}
extern "C" int gl_circuit_description(const gl_circuit* c, gl_circuit_desc* out) {
    GL_REQUIRE(c && out, GL_ERR_ARG, "null argument");
    *out = c->desc;
    return GL_OK;
}
extern "C" int gl_circuit_digest(const gl_circuit* c, uint64_t h_out[4]) {
    GL_REQUIRE(c && h_out, GL_ERR_ARG, "null argument");
    memcpy(h_out, c->circuit_digest, 32);
    return GL_OK;
}
extern "C" int gl_circuit_constants_sigmas_cap(const gl_circuit* c, uint64_t* h_out) {
    GL_REQUIRE(c && h_out, GL_ERR_ARG, "null argument");
    return gl_batch_cap(c->cs_batch, h_out);
}
extern "C" const gl_batch* gl_circuit_constants_sigmas_batch(const gl_circuit* c) { return c ? c->cs_batch : nullptr; }

// ---- helpers -------------------------------------------------------------------------------------------------------------
static inline gl2_t h_ext(gl_t a, gl_t b) { return gl2_make(a, b); }
static void put_u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_words(std::vector<uint8_t>& o, const gl_t* v, size_t n) { for (size_t i = 0; i < n; i++) put_u64(o, gl_canon(v[i])); }
static uint32_t host_bitrev32(uint32_t x, uint32_t bits) { uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) r = (r << 1) | ((x >> i) & 1); return r; }

struct BatchHolder { gl_batch* b = nullptr; ~BatchHolder() { if (b) gl_batch_free(b); } };
struct MerkleHolder { gl_ctx* c; GlMerkle m; explicit MerkleHolder(gl_ctx* ctx) : c(ctx) {} ~MerkleHolder() { gl_merkle_release(c, &m); } };

static int d2h(gl_ctx* c, void* dst, const void* src, size_t bytes) { return gl_copy_d2h(c, dst, src, bytes); }
static int h2d_async(gl_ctx* c, void* dst, const void* src, size_t bytes) {
    // sources are small host vectors that outlive the next sync; pageable H2D is staged by the runtime before returning
    GL_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return GL_OK;
}

// Host witness -> HBM through a two-deep ring of pinned chunks (the drop-in entry points gl_prove / gl_prove_columns; reference side:
// MatrixWitness.wire_values, iop/witness.rs:256-258, handed to PolynomialBatch::from_values at plonk/prover.rs:145-156).
// A hipMemcpyAsync from PAGEABLE memory is staged by the runtime inside the call: the calling thread spins there until the
// stream's earlier work is done and the copy is on its way (the round-2 finding for the D2H direction: a full core per proof in
// flight).  Here the host thread copies the columns into one of two pinned chunks (plain memcpy, ~10 GB/s), hands the chunk to
// the copy engine and fills the other one meanwhile; it only ever waits, sleeping between polls, for a chunk to come back.
// The DMA of a proof's witness runs under the kernels of the other proofs in flight on the device's other contexts.
static hipError_t gl_event_wait(hipEvent_t ev) {
    hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(30)) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else { struct timespec ts = {0, 20000}; (void)nanosleep(&ts, nullptr); }
    }
}
struct H2dRing {
    gl_ctx* c; void* buf[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
    explicit H2dRing(gl_ctx* ctx) : c(ctx) {}
    int init(size_t chunk_bytes) {
        for (int k = 0; k < 2; k++) {
            GL_TRY(c->pin_acquire(chunk_bytes, &buf[k], &cap[k]));
            GL_CHECK_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        return GL_OK;
    }
    int slot_free(int k) { if (busy[k]) { GL_CHECK_HIP(gl_event_wait(ev[k])); busy[k] = false; } return GL_OK; }
    int send(int k, void* d_dst, size_t bytes) {
        GL_CHECK_HIP(hipMemcpyAsync(d_dst, buf[k], bytes, hipMemcpyHostToDevice, c->stream));
        GL_CHECK_HIP(hipEventRecord(ev[k], c->stream));
        busy[k] = true;
        return GL_OK;
    }
    ~H2dRing() {                  // a chunk goes back to the context's list only after the copy engine has read it
        for (int k = 0; k < 2; k++) {
            if (busy[k]) (void)gl_event_wait(ev[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (buf[k]) c->pin_release(buf[k], cap[k]);
        }
    }
};
static const size_t GL_H2D_CHUNK = size_t(8) << 20;
// cols != null: ncols host vectors of n elements each (d_dst[c * n + i] = cols[c][i]); else `flat`, ncols * n contiguous elements
static int h2d_witness(gl_ctx* ctx, gl_t* d_dst, size_t ncols, size_t n, const uint64_t* const* cols, const uint64_t* flat) {
    const size_t col_bytes = n * sizeof(gl_t), total = ncols * col_bytes;
    if (!total) return GL_OK;
    H2dRing ring(ctx);
    GL_TRY(ring.init(total < GL_H2D_CHUNK ? total : GL_H2D_CHUNK));
    const size_t chunk = ring.cap[0] < ring.cap[1] ? ring.cap[0] : ring.cap[1];
    size_t off = 0; int k = 0;
    while (off < total) {
        const size_t len = total - off < chunk ? total - off : chunk;
        GL_TRY(ring.slot_free(k));
        if (!cols) memcpy(ring.buf[k], (const char*)flat + off, len);
        else {
            size_t done = 0;
            while (done < len) {                        // a chunk may begin and end inside a column
                const size_t c = (off + done) / col_bytes, in_col = (off + done) % col_bytes;
                const size_t piece = (col_bytes - in_col) < (len - done) ? (col_bytes - in_col) : (len - done);
                memcpy((char*)ring.buf[k] + done, (const char*)cols[c] + in_col, piece);
                done += piece;
            }
        }
        GL_TRY(ring.send(k, (char*)d_dst + off, len));
        off += len; k ^= 1;
    }
    return GL_OK;                                       // ~H2dRing waits for the last two chunks (the DMA of <= 16 MiB)
}

// ======================================================================================================================
// Phase-level entry points: the seam of SURVEY 8(b).  A caller that keeps the Fiat-Shamir transcript on its side (the
// reference's Challenger in Rust) drives these one by one; gl_prove() below is exactly that driver with the transcript
// in C++.  Every phase leaves its polynomials in HBM and hands back only what the transcript needs.
// ======================================================================================================================

// the coset shifts of get_unique_coset_shifts (field/src/cosets.rs:9-24) are 1, 7, 7^2, ...: lets the kernels step beta x k_j by x7
static uint32_t k_is_are_powers_of_7(const gl_circuit_desc& d) {
    gl_t x = 1;
    for (int j = 0; j < 80; j++) { if (gl_canon(d.k_is[j]) != x) return 0; x = gl_canon(gl_mul(x, GL_MULT_GENERATOR)); }
    return 1;
}

// ---- 6. all_wires_permutation_partial_products (plonk/prover.rs:332-416): d_zs[20][n] VALUES ----
static int partial_products_values(gl_ctx* ctx, const gl_circuit* cir, const gl_t* d_wires, const gl_t* betas, const gl_t* gammas, gl_t* d_zs) {
    const gl_circuit_desc& d = cir->desc;
    const size_t n = cir->n;
    hipStream_t st = ctx->stream;
    DevBuf d_chunk(ctx), d_rowp(ctx), d_seg(ctx);
    const uint32_t nseg = (uint32_t)((n + GLP_SEG - 1) / GLP_SEG);
    GL_TRY(d_chunk.alloc(2 * GLP_CHUNKS * n * sizeof(gl_t)));
    DevBuf d_den(ctx); GL_TRY(d_den.alloc(2 * GLP_CHUNKS * n * sizeof(gl_t)));
    GL_TRY(d_rowp.alloc(2 * n * sizeof(gl_t)));
    GL_TRY(d_seg.alloc(2 * (size_t)nseg * sizeof(gl_t)));
    GlPowTable xt;
    GL_TRY(ctx->get_pow_table(gl_host_root_of_unity(d.degree_bits), 1, (uint32_t)((n + 2047) >> 11), &xt));
    GlPermParams pp;
    pp.wires = d_wires; pp.sigmas = cir->d_sigmas; pp.xpow_lo = xt.lo; pp.xpow_hi = xt.hi;
    for (int j = 0; j < 80; j++) pp.k_is[j] = d.k_is[j];
    pp.k_is_powers_of_7 = k_is_are_powers_of_7(d);
    for (int i = 0; i < 2; i++) { pp.betas[i] = gl_canon(betas[i]); pp.gammas[i] = gl_canon(gammas[i]); }
    pp.n = (uint32_t)n; pp.chunk_prod = d_chunk.as<gl_t>(); pp.row_prod = d_rowp.as<gl_t>();
    ctx->timing_begin("compute partial products");
    hipLaunchKernelGGL(k_pp_chunk_terms, dim3((unsigned)((n + 255) / 256), 2 * GLP_CHUNKS), dim3(256), 0, st, pp, d_den.as<gl_t>());
    hipLaunchKernelGGL(k_pp_chunk_products, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, st, pp, d_den.as<const gl_t>());
    hipLaunchKernelGGL(k_z_segment_products, dim3(nseg, 2), dim3(256), 0, st, d_rowp.as<gl_t>(), (uint32_t)n, d_seg.as<gl_t>());
    hipLaunchKernelGGL(k_z_segment_scan, dim3(1), dim3(64), 0, st, d_seg.as<gl_t>(), nseg);
    hipLaunchKernelGGL(k_z_finalize, dim3(nseg, 2), dim3(256), 0, st, d_rowp.as<gl_t>(), d_chunk.as<gl_t>(), d_seg.as<gl_t>(), (uint32_t)n, d_zs);
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
// ---- compute_all_lookup_polys (plonk/prover.rs:425-572): d_out[2 * 7][n] VALUES (RE + 6 partial SLDC per challenge) ----
static int lookup_polys_values(gl_ctx* ctx, const gl_circuit* cir, const gl_t* d_wires, const gl_t* deltas8, gl_t* d_out) {
    const gl_circuit_desc& d = cir->desc;
    const size_t n = cir->n;
    hipStream_t st = ctx->stream;
    uint32_t max_rows = 0;
    for (unsigned t = 0; t < d.num_luts; t++) max_rows = std::max(max_rows, d.first_lut_row[t] - d.last_lu_row[t] + 1);
    DevBuf d_agg(ctx); GL_TRY(d_agg.alloc((size_t)2 * max_rows * 8 * sizeof(gl_t)));
    gl_t dl[8]; for (int i = 0; i < 8; i++) dl[i] = gl_canon(deltas8[i]);
    ctx->timing_begin("compute lookup polys");
    GL_CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)2 * d.num_lookup_polys * n * sizeof(gl_t), st));
    // the tables share the polynomials and own disjoint row ranges, each starting from zero on the row above its first table row
    // (prover.rs:449-454: one pass per LookupWire); the launches of one table follow the previous table's on the stream
    for (unsigned t = 0; t < d.num_luts; t++) {
        const uint32_t nrows = d.first_lut_row[t] - d.last_lu_row[t] + 1;
        hipLaunchKernelGGL(k_lookup_inverses, dim3(nrows, 2), dim3(64), 0, st, d_wires, (uint32_t)n, d.last_lu_row[t], d.last_lut_row[t],
                           dl[glhost::LU_CH_A], dl[glhost::LU_CH_ALPHA], dl[glhost::LU_CH_B], dl[glhost::LU_CH_DELTA],
                           dl[4 + glhost::LU_CH_A], dl[4 + glhost::LU_CH_ALPHA], dl[4 + glhost::LU_CH_B], dl[4 + glhost::LU_CH_DELTA], d_agg.as<gl_t>());
        hipLaunchKernelGGL(k_lookup_scan, dim3(1), dim3(64), 0, st, (uint32_t)n, d.last_lu_row[t], d.last_lut_row[t], d.first_lut_row[t],
                           dl[glhost::LU_CH_DELTA], dl[4 + glhost::LU_CH_DELTA], d_agg.as<const gl_t>(), d_out);
    }
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    return GL_OK;
}
static int check_phase_args(gl_ctx* ctx, const gl_circuit* cir) {
    GL_REQUIRE(ctx && cir, GL_ERR_ARG, "null context / circuit");
    GL_REQUIRE(cir->ctx->device == ctx->device, GL_ERR_ARG, "circuit lives on another device");
    return ctx->activate();
}
static int check_batch(const gl_circuit* cir, const gl_batch* b, size_t ncols, const char* what) {
    GL_REQUIRE(b && b->ncols == ncols && b->n == cir->n && b->rate_bits == cir->desc.rate_bits && b->cap_height == cir->desc.cap_height, GL_ERR_ARG, what);
    return GL_OK;
}
static int partial_products_phase(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* d_wires, const uint64_t* betas, const uint64_t* gammas, const uint64_t* deltas8, gl_batch** out) {
    GL_TRY(check_phase_args(ctx, cir));
    GL_REQUIRE(d_wires && betas && gammas && out, GL_ERR_ARG, "gl_partial_products: null argument");
    const size_t nlk = 2 * (size_t)cir->desc.num_lookup_polys, nzs = 20 + nlk;
    GL_REQUIRE((nlk != 0) == (deltas8 != nullptr), GL_ERR_ARG, "circuits with lookups take gl_partial_products_lookups (with the delta challenges), circuits without take gl_partial_products");
    DevBuf d_zs(ctx); GL_TRY(d_zs.alloc(nzs * cir->n * sizeof(gl_t)));
    GL_TRY(partial_products_values(ctx, cir, d_wires, betas, gammas, d_zs.as<gl_t>()));
    if (nlk) GL_TRY(lookup_polys_values(ctx, cir, d_wires, deltas8, d_zs.as<gl_t>() + 20 * cir->n));
    return gl_batch_from_device(ctx, d_zs.as<uint64_t>(), nzs, cir->n, cir->desc.rate_bits, cir->desc.cap_height, 1, out);
}
extern "C" int gl_partial_products(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* d_wires, const uint64_t betas[2], const uint64_t gammas[2], gl_batch** out) {
    return partial_products_phase(ctx, cir, d_wires, betas, gammas, nullptr, out);
}
extern "C" int gl_partial_products_lookups(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* d_wires, const uint64_t betas[2], const uint64_t gammas[2],
                                           const uint64_t deltas[8], gl_batch** out) {
    GL_REQUIRE(deltas, GL_ERR_ARG, "gl_partial_products_lookups: null deltas");
    return partial_products_phase(ctx, cir, d_wires, betas, gammas, deltas, out);
}

// ---- 9. compute_quotient_polys + split (plonk/prover.rs:229-258,574-744): d_q[2][8n] -> the 16 chunk COEFFICIENT columns ----
static int quotient_chunks(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* wires, const gl_batch* zs, const gl_t* pi_hash,
                           const gl_t* betas, const gl_t* gammas, const gl_t* alphas, gl_t* d_q, std::vector<gl_t>& apow, const gl_t* deltas8 = nullptr) {
    const gl_circuit_desc& d = cir->desc;
    const size_t n = cir->n, N = n << d.rate_bits;
    const uint32_t lgn = d.degree_bits, lgN = lgn + d.rate_bits;
    hipStream_t st = ctx->stream;
    apow.assign(2 * GLQ_MAX_TERMS, 0);
    for (int b = 0; b < 2; b++) { gl_t x = 1; const gl_t al = gl_canon(alphas[b]); for (int t = 0; t < GLQ_MAX_TERMS; t++) { apow[b * GLQ_MAX_TERMS + t] = x; x = gl_canon(gl_mul(x, al)); } }
    GL_TRY(ctx->ensure_dev_small(1 << 20));
    gl_t* d_apow = ctx->dev_small;
    GL_TRY(h2d_async(ctx, d_apow, apow.data(), apow.size() * sizeof(gl_t)));
    GlPowTable xt;
    GL_TRY(ctx->get_pow_table(gl_host_root_of_unity(lgN), GL_MULT_GENERATOR, (uint32_t)((N + 2047) >> 11), &xt));
    GlQuotParams q;
    ::memset((void*)&q, 0, sizeof q);
    q.cs = cir->cs_batch->lde; q.wires = wires->lde; q.zs = zs->lde; q.xpow_lo = xt.lo; q.xpow_hi = xt.hi;
    q.alpha_pows = d_apow; q.out = d_q;
    for (int j = 0; j < 80; j++) q.k_is[j] = d.k_is[j];
    q.k_is_powers_of_7 = k_is_are_powers_of_7(d);
    for (int i = 0; i < 2; i++) { q.betas[i] = gl_canon(betas[i]); q.gammas[i] = gl_canon(gammas[i]); }
    for (int i = 0; i < 4; i++) q.pi_hash[i] = gl_canon(pi_hash[i]);
    {   // ZeroPolyOnCoset (field/src/zero_poly_coset.rs:19-36)
        gl_t g_pow_n = GL_MULT_GENERATOR; for (uint32_t i = 0; i < lgn; i++) g_pow_n = gl_sqr(g_pow_n);
        gl_t w8 = gl_host_root_of_unity(d.rate_bits), x = 1;
        for (int i = 0; i < 8; i++) { q.zh_evals[i] = gl_canon(gl_sub(gl_mul(g_pow_n, x), 1)); q.zh_inv[i] = gl_canon(gl_inv(q.zh_evals[i])); x = gl_mul(x, w8); }
    }
    q.l0_coset = cir->d_l0_coset;
    q.n_field = (gl_t)n; q.lgN = lgN; q.num_constants = d.num_constants; q.num_selectors = d.num_selectors; q.num_gates = d.num_gates;
    q.next_step = 1u << d.rate_bits;
    q.num_lookup_selectors = d.num_lookup_selectors; q.num_lookup_polys = d.num_lookup_polys;
    q.num_luts = d.num_luts;
    q.gate_term0 = 2 + 2 * GLP_CHUNKS + (d.num_lookup_polys ? 2 * GLQ_LOOKUP_TERMS(d.num_luts) : 0);
    if (d.num_lookup_polys) {
        GL_REQUIRE(deltas8, GL_ERR_ARG, "a circuit with lookups needs the delta challenges");
        for (int i = 0; i < 8; i++) q.deltas[i] = gl_canon(deltas8[i]);
        for (int c = 0; c < 2; c++)
            for (unsigned t = 0; t < d.num_luts; t++)
                q.lut_poly_at_delta[c][t] = glhost::lut_poly_at_delta(d, t, q.deltas[4 * c + glhost::LU_CH_B], q.deltas[4 * c + glhost::LU_CH_DELTA]);
    }
    for (unsigned g = 0; g < d.num_gates; g++) { q.gate_types[g] = d.gate_types[g]; q.gate_params[g] = d.gate_params[g]; q.gate_sel[g] = d.gate_selector_index[g]; q.group_start[g] = d.gate_group_start[g]; q.group_end[g] = d.gate_group_end[g]; }
    ctx->timing_begin("compute quotient polys");
    hipLaunchKernelGGL(k_quotient<false>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, q);
    bool has_poseidon_gate = false;
    for (unsigned g = 0; g < d.num_gates; g++) has_poseidon_gate |= d.gate_types[g] == 4;
    if (has_poseidon_gate) hipLaunchKernelGGL(k_quotient<true>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, q);
    bool has_random_access_gate = false;
    for (unsigned g = 0; g < d.num_gates; g++) has_random_access_gate |= d.gate_types[g] == glhost::G_RANDOM_ACCESS;
    if (has_random_access_gate) hipLaunchKernelGGL(k_quotient_random_access, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, q);
    if (d.num_lookup_polys) hipLaunchKernelGGL(k_quotient_lookup, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, q);
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    // coset_ifft(7) of each quotient (prover.rs:739-743); the 8n coefficients ARE the 8 chunks of n (prover.rs:245-258)
    return gl_ntt_run(ctx, d_q, N, (uint32_t)N, d_q, N, lgN, 2, true, 0, gl_canon(gl_inv(GL_MULT_GENERATOR)), gl_host_inverse_2exp(lgN));
}
static int quotient_phase(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* wires, const gl_batch* zs_partial_products, const uint64_t* pi_hash,
                          const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas, const uint64_t* deltas8, gl_batch** out);
extern "C" int gl_quotient_polys(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* wires, const gl_batch* zs_partial_products, const uint64_t pi_hash[4],
                                 const uint64_t betas[2], const uint64_t gammas[2], const uint64_t alphas[2], gl_batch** out) {
    return quotient_phase(ctx, cir, wires, zs_partial_products, pi_hash, betas, gammas, alphas, nullptr, out);
}
extern "C" int gl_quotient_polys_lookups(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* wires, const gl_batch* zs_partial_products_lookups, const uint64_t pi_hash[4],
                                         const uint64_t betas[2], const uint64_t gammas[2], const uint64_t alphas[2], const uint64_t deltas[8], gl_batch** out) {
    GL_REQUIRE(deltas, GL_ERR_ARG, "gl_quotient_polys_lookups: null deltas");
    return quotient_phase(ctx, cir, wires, zs_partial_products_lookups, pi_hash, betas, gammas, alphas, deltas, out);
}
static int quotient_phase(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* wires, const gl_batch* zs_partial_products, const uint64_t* pi_hash,
                          const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas, const uint64_t* deltas8, gl_batch** out) {
    GL_TRY(check_phase_args(ctx, cir));
    GL_REQUIRE(pi_hash && betas && gammas && alphas && out, GL_ERR_ARG, "gl_quotient_polys: null argument");
    const size_t nlk = 2 * (size_t)cir->desc.num_lookup_polys;
    GL_REQUIRE((nlk != 0) == (deltas8 != nullptr), GL_ERR_ARG, "circuits with lookups take gl_quotient_polys_lookups (with the delta challenges), circuits without take gl_quotient_polys");
    GL_TRY(check_batch(cir, wires, 135, "gl_quotient_polys: wires batch does not match the circuit"));
    GL_TRY(check_batch(cir, zs_partial_products, 20 + nlk, "gl_quotient_polys: Z / partial-products (/ lookups) batch does not match the circuit"));
    const size_t N = cir->n << cir->desc.rate_bits;
    DevBuf d_q(ctx); GL_TRY(d_q.alloc(2 * N * sizeof(gl_t)));
    std::vector<gl_t> apow;
    GL_TRY(quotient_chunks(ctx, cir, wires, zs_partial_products, pi_hash, betas, gammas, alphas, d_q.as<gl_t>(), apow, deltas8));
    int rc = gl_batch_from_device(ctx, d_q.as<uint64_t>(), 16, cir->n, cir->desc.rate_bits, cir->desc.cap_height, 0, out);
    GL_CHECK_HIP(gl_stream_wait(ctx->stream));      // `apow` was the source of an async upload
    return rc;
}

// ---- 12. OpeningSet::new (plonk/proof.rs:306-344): polynomials `first .. first + count` of a batch at an extension point ----
static void launch_open(gl_ctx* ctx, const gl_batch* b, size_t first, size_t count, gl2_t z, gl_t* d_out) {
    hipLaunchKernelGGL(k_eval_at_ext, dim3((unsigned)count), dim3(256), 0, ctx->stream, b->coeffs + first * b->n, (uint32_t)b->n, (uint64_t)b->n, z.a, z.b, d_out);
}
extern "C" int gl_open_at(gl_ctx* ctx, const gl_batch* b, const uint64_t z[2], size_t first_col, size_t num_cols, uint64_t* h_out) {
    GL_REQUIRE(ctx && b && z && h_out, GL_ERR_ARG, "gl_open_at: null argument");
    GL_REQUIRE(first_col <= b->ncols && num_cols <= b->ncols - first_col && num_cols <= 4096, GL_ERR_ARG, "gl_open_at: column range out of bounds");
    if (!num_cols) return GL_OK;
    GL_TRY(ctx->activate());
    GL_TRY(ctx->ensure_dev_small(1 << 20));
    gl_t* d_open = ctx->dev_small + 4096;
    ctx->timing_begin("construct the opening set");
    launch_open(ctx, b, first_col, num_cols, gl2_make(gl_canon(z[0]), gl_canon(z[1])), d_open);
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    return d2h(ctx, h_out, d_open, 2 * num_cols * sizeof(gl_t));
}

// ---- 14. PolynomialBatch::prove_openings (fri/oracle.rs:162-219) + fri_proof (fri/prover.rs:20-216), split at every
//          transcript dependency ----
struct gl_fri {
    gl_ctx* ctx = nullptr;
    gl_circuit_desc desc;
    const gl_batch* oracles[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t n = 0;
    uint32_t lgN = 0;
    // commit phase state
    unsigned round = 0;                      // rounds folded so far
    bool committed = false;                  // the current round's tree exists and waits for its beta
    std::unique_ptr<DevBuf> cur_vals;        // value planes [2][2^cur_lgN] of the current codeword
    std::unique_ptr<DevBuf> coef;            // coefficient planes [2][cur_n]
    size_t cur_n = 0; uint32_t cur_lgN = 0; gl_t shift = GL_MULT_GENERATOR;
    std::vector<std::unique_ptr<MerkleHolder>> trees;
    std::vector<std::unique_ptr<DevBuf>> vals;           // value planes of each committed round (query phase)
    std::vector<uint32_t> lg;
    // host sources of asynchronous uploads (alive until the object dies)
    std::vector<const gl_t*> h_cols; std::vector<gl_t> h_apow;
};
extern "C" void gl_fri_free(gl_fri* f) {
    if (!f) return;
    gl_ctx* ctx = f->ctx;
    if (ctx) (void)gl_stream_wait(ctx->stream);
    delete f;                                  // its trees and buffers go back to the context's pool first
    gl_ctx_release(ctx);
}
extern "C" int gl_fri_combine(gl_ctx* ctx, const gl_circuit* cir, const gl_batch* const batches[4], const uint64_t zeta_in[2], const uint64_t alpha_in[2], gl_fri** out) {
    GL_TRY(check_phase_args(ctx, cir));
    GL_REQUIRE(batches && zeta_in && alpha_in && out, GL_ERR_ARG, "gl_fri_combine: null argument");
    const gl_circuit_desc& d = cir->desc;
    const size_t n = cir->n, N = n << d.rate_bits;
    const size_t nlk = 2 * (size_t)d.num_lookup_polys;                              // lookup polynomials: behind Z||partial products in oracle 2
    const size_t ncs = d.num_constants + 80, nopen = ncs + 135 + 20 + 16 + nlk, nnext = 2 + nlk;
    const size_t want[4] = {ncs, 135, 20 + nlk, 16};
    for (int o = 0; o < 4; o++) GL_TRY(check_batch(cir, batches[o], want[o], "gl_fri_combine: oracle order is constants||sigmas, wires, Z||partial products(||lookups), quotient"));
    hipStream_t st = ctx->stream;
    std::unique_ptr<gl_fri, void (*)(gl_fri*)> f(new gl_fri(), gl_fri_free);
    f->ctx = ctx; ctx->retain(); f->desc = d; f->n = n; f->lgN = d.degree_bits + d.rate_bits;
    for (int o = 0; o < 4; o++) f->oracles[o] = batches[o];
    const gl2_t zeta = gl2_make(gl_canon(zeta_in[0]), gl_canon(zeta_in[1])), fri_alpha = gl2_make(gl_canon(alpha_in[0]), gl_canon(alpha_in[1]));
    const gl2_t gzeta = gl2_canon(gl2_scalar(zeta, gl_host_root_of_unity(d.degree_bits)));
    f->coef.reset(new DevBuf(ctx)); GL_TRY(f->coef->alloc(2 * n * sizeof(gl_t)));            // planes a, b of alpha^2 Q0 + Q1
    {
        DevBuf d_F(ctx), d_heads(ctx), d_cols(ctx), d_apow(ctx);
        const uint32_t seg_len = (uint32_t)(n / 1024 > 32 ? n / 1024 : (n >= 32 ? 32 : n));      // at most 1024 segments
        const uint32_t nseg = (uint32_t)((n + seg_len - 1) / seg_len);
        GL_TRY(d_F.alloc(2 * n * sizeof(gl_t)));
        GL_TRY(d_heads.alloc(2 * (size_t)nseg * sizeof(gl_t)));
        GL_TRY(d_cols.alloc((nopen + nnext) * sizeof(gl_t*)));
        GL_TRY(d_apow.alloc(2 * (nopen + nnext) * sizeof(gl_t)));
        std::vector<const gl_t*>& cols = f->h_cols;
        // fri_all_polys / fri_next_batch_polys (circuit_data.rs:564-597): the lookup polynomials come LAST in both batches
        for (int o = 0; o < 4; o++) for (size_t c = 0; c < (o == 2 ? (size_t)20 : batches[o]->ncols); c++) cols.push_back(batches[o]->coeffs + c * n);
        for (size_t c = 0; c < nlk; c++) cols.push_back(batches[2]->coeffs + (20 + c) * n);
        cols.push_back(batches[2]->coeffs); cols.push_back(batches[2]->coeffs + n);
        for (size_t c = 0; c < nlk; c++) cols.push_back(batches[2]->coeffs + (20 + c) * n);
        std::vector<gl_t>& apow = f->h_apow; apow.assign(2 * (nopen + nnext), 0);
        { gl2_t x = gl2_make(1, 0); for (size_t j = 0; j < nopen; j++) { apow[2 * j] = x.a; apow[2 * j + 1] = x.b; x = gl2_canon(gl2_mul(x, fri_alpha)); } }
        gl2_t shift = gl2_make(1, 0);     // alpha^(#polys of batch 1) (reducing.rs:103-106)
        { gl2_t x = gl2_make(1, 0); for (size_t j = 0; j < nnext; j++) { apow[2 * (nopen + j)] = x.a; apow[2 * (nopen + j) + 1] = x.b; x = gl2_canon(gl2_mul(x, fri_alpha)); } shift = x; }
        GL_TRY(h2d_async(ctx, d_cols.p, cols.data(), cols.size() * sizeof(gl_t*)));
        GL_TRY(h2d_async(ctx, d_apow.p, apow.data(), apow.size() * sizeof(gl_t)));
        gl_t* Fa = d_F.as<gl_t>(); gl_t* Fb = Fa + n;
        gl_t* Qa = f->coef->as<gl_t>(); gl_t* Qb = Qa + n;
        const unsigned gb = (unsigned)((n + 63) / 64), sb = (nseg + 63) / 64;      // k_fri_combine: 64 coefficients per workgroup
        ctx->timing_begin("reduce batch + divide by linear");
        // batch 0: all polynomials at zeta
        hipLaunchKernelGGL(k_fri_combine, dim3(gb), dim3(256), 0, st, d_cols.as<const gl_t*>(), d_apow.as<gl_t>(), (uint32_t)nopen, (uint32_t)n, Fa, Fb);
        hipLaunchKernelGGL(k_div_linear_heads, dim3(sb), dim3(64), 0, st, Fa, Fb, (uint32_t)n, seg_len, zeta.a, zeta.b, d_heads.as<gl_t>());
        hipLaunchKernelGGL(k_div_linear_carries, dim3(1), dim3(1024), 0, st, d_heads.as<gl_t>(), (uint32_t)n, seg_len, zeta.a, zeta.b);
        hipLaunchKernelGGL(k_div_linear_apply, dim3(sb), dim3(64), 0, st, Fa, Fb, (uint32_t)n, seg_len, zeta.a, zeta.b, d_heads.as<gl_t>(), shift.a, shift.b, Qa, Qb, 0);
        // batch 1: the Z polynomials at g * zeta
        hipLaunchKernelGGL(k_fri_combine, dim3(gb), dim3(256), 0, st, d_cols.as<const gl_t*>() + nopen, d_apow.as<gl_t>() + 2 * nopen, (uint32_t)nnext, (uint32_t)n, Fa, Fb);
        hipLaunchKernelGGL(k_div_linear_heads, dim3(sb), dim3(64), 0, st, Fa, Fb, (uint32_t)n, seg_len, gzeta.a, gzeta.b, d_heads.as<gl_t>());
        hipLaunchKernelGGL(k_div_linear_carries, dim3(1), dim3(1024), 0, st, d_heads.as<gl_t>(), (uint32_t)n, seg_len, gzeta.a, gzeta.b);
        hipLaunchKernelGGL(k_div_linear_apply, dim3(sb), dim3(64), 0, st, Fa, Fb, (uint32_t)n, seg_len, gzeta.a, gzeta.b, d_heads.as<gl_t>(), (gl_t)1, (gl_t)0, Qa, Qb, 1);
        ctx->timing_end();
        GL_CHECK_HIP(hipGetLastError());
    }
    // final_poly.lde(rate_bits).coset_fft(7) on both planes (fri/oracle.rs:199-204)
    f->cur_vals.reset(new DevBuf(ctx)); GL_TRY(f->cur_vals->alloc(2 * N * sizeof(gl_t)));
    GL_TRY(gl_ntt_run(ctx, f->coef->as<gl_t>(), n, (uint32_t)n, f->cur_vals->as<gl_t>(), N, f->lgN, 2, false, GL_MULT_GENERATOR, 0, 1));
    f->cur_n = n; f->cur_lgN = f->lgN;
    *out = f.release();
    return GL_OK;
}
// fri_committed_trees, first half of one loop iteration (fri/prover.rs:76-92): Merkle tree of the current codeword
extern "C" int gl_fri_commit_round(gl_fri* f, uint64_t* h_cap_out) {
    GL_REQUIRE(f && h_cap_out, GL_ERR_ARG, "gl_fri_commit_round: null argument");
    GL_REQUIRE(f->round < f->desc.num_fri_rounds && !f->committed, GL_ERR_ARG, "gl_fri_commit_round: no round left to commit (call gl_fri_fold first)");
    gl_ctx* ctx = f->ctx;
    GL_TRY(ctx->activate());
    const uint32_t ab = f->desc.fri_arity_bits[f->round], arity = 1u << ab;
    const size_t curN = size_t(1) << f->cur_lgN;
    // leaves: `arity` consecutive entries of the bit-reversed value array, flattened (fri/prover.rs:80-89)
    std::vector<uint64_t> offs(2 * arity);
    for (uint32_t k = 0; k < arity; k++)
        for (uint32_t cpt = 0; cpt < 2; cpt++) offs[2 * k + cpt] = (uint64_t)cpt * curN + (uint64_t)host_bitrev32(k, ab) * (curN >> ab);
    std::unique_ptr<MerkleHolder> tree(new MerkleHolder(ctx));
    GL_TRY(gl_merkle_build(ctx, f->cur_vals->as<gl_t>(), offs.data(), 2 * arity, f->cur_lgN - ab, f->desc.cap_height, &tree->m));
    GL_TRY(d2h(ctx, h_cap_out, tree->m.level_ptr(tree->m.num_levels() - 1), (size_t(4) << f->desc.cap_height) * sizeof(gl_t)));
    f->trees.push_back(std::move(tree));
    f->committed = true;
    return GL_OK;
}
// second half (fri/prover.rs:94-103): fold the coefficients by beta, next codeword on the coset shift^arity
extern "C" int gl_fri_fold(gl_fri* f, const uint64_t beta_in[2]) {
    GL_REQUIRE(f && beta_in, GL_ERR_ARG, "gl_fri_fold: null argument");
    GL_REQUIRE(f->committed, GL_ERR_ARG, "gl_fri_fold: commit the round first");
    gl_ctx* ctx = f->ctx;
    GL_TRY(ctx->activate());
    const uint32_t ab = f->desc.fri_arity_bits[f->round], arity = 1u << ab;
    const gl2_t beta = gl2_make(gl_canon(beta_in[0]), gl_canon(beta_in[1]));
    const size_t next_n = f->cur_n >> ab;
    GL_REQUIRE(next_n >= 1, GL_ERR_INTERNAL, "FRI fold below one coefficient");
    std::unique_ptr<DevBuf> next(new DevBuf(ctx));
    GL_TRY(next->alloc(2 * next_n * sizeof(gl_t)));
    ctx->timing_begin("fold codewords in the commitment phase");
    hipLaunchKernelGGL(k_fri_fold, dim3((unsigned)((next_n + 255) / 256)), dim3(256), 0, ctx->stream, f->coef->as<gl_t>(), f->coef->as<gl_t>() + f->cur_n,
                       (uint32_t)next_n, arity, beta.a, beta.b, next->as<gl_t>(), next->as<gl_t>() + next_n);
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    f->shift = gl_canon(gl_exp(f->shift, arity));
    f->vals.push_back(std::move(f->cur_vals)); f->lg.push_back(f->cur_lgN);
    f->cur_lgN -= ab;
    f->cur_vals.reset(new DevBuf(ctx));
    GL_TRY(f->cur_vals->alloc(2 * (size_t(1) << f->cur_lgN) * sizeof(gl_t)));
    GL_TRY(gl_ntt_run(ctx, next->as<gl_t>(), next_n, (uint32_t)next_n, f->cur_vals->as<gl_t>(), size_t(1) << f->cur_lgN, f->cur_lgN, 2, false, f->shift, 0, 1));
    f->coef = std::move(next);
    f->cur_n = next_n;
    f->round++; f->committed = false;
    return GL_OK;
}
// final polynomial: the remaining non-zero coefficients (coeffs.truncate(len >> rate_bits), fri/prover.rs:106-111), interleaved (a, b)
extern "C" int gl_fri_final_poly(gl_fri* f, uint64_t* h_out, size_t cap_words, size_t* num_words) {
    GL_REQUIRE(f && num_words, GL_ERR_ARG, "gl_fri_final_poly: null argument");
    GL_REQUIRE(f->round == f->desc.num_fri_rounds && !f->committed, GL_ERR_ARG, "gl_fri_final_poly: reduction rounds not finished");
    *num_words = 2 * f->cur_n;
    if (!h_out) return GL_OK;
    GL_REQUIRE(cap_words >= 2 * f->cur_n, GL_ERR_ARG, "gl_fri_final_poly: output too small");
    GL_TRY(f->ctx->activate());
    std::vector<gl_t> fin(2 * f->cur_n);
    GL_TRY(d2h(f->ctx, fin.data(), f->coef->p, 2 * f->cur_n * sizeof(gl_t)));
    for (size_t i = 0; i < f->cur_n; i++) { h_out[2 * i] = fin[i]; h_out[2 * i + 1] = fin[f->cur_n + i]; }
    return GL_OK;
}

// ---- fri_proof_of_work (fri/prover.rs:115-160): smallest w such that permute(state with the pending inputs and w)[7] has
//      enough leading zeros.  `sponge_state` is the Challenger's sponge, `input_buffer[0..input_len)` its pending inputs. ----
extern "C" int gl_pow_grind(gl_ctx* ctx, const uint64_t sponge_state[12], const uint64_t* input_buffer, uint32_t input_len, uint32_t min_leading_zeros, uint64_t* witness) {
    GL_REQUIRE(ctx && sponge_state && witness && (input_buffer || !input_len), GL_ERR_ARG, "gl_pow_grind: null argument");
    GL_REQUIRE(input_len < 8 && min_leading_zeros <= 40, GL_ERR_ARG, "gl_pow_grind: the witness must fit the rate (input_len < 8), at most 40 bits of work");
    GL_TRY(ctx->activate());
    hipStream_t st = ctx->stream;
    GL_TRY(ctx->ensure_dev_small(1 << 20));
    unsigned long long* d_res = (unsigned long long*)ctx->dev_small;
    GlPowParams pw;
    for (int i = 0; i < 12; i++) pw.state[i] = sponge_state[i];
    for (uint32_t i = 0; i < input_len; i++) pw.state[i] = input_buffer[i];
    pw.pos = input_len; pw.min_leading_zeros = min_leading_zeros; pw.result = d_res;
    // (measured, 16 proofs in flight: windows of 2^pow_bits, and a small grid that walks a longer window in ascending sweeps and
    // stops at the first witness -- 40 % fewer permutations -- gave the same 287 proofs/s within noise, resp. 8 % less for
    // sweeps of 2^15 lanes whose long low-occupancy launches hold up their proof; the grind is 2.7 % of a proof's instructions)
    // expected 2^pow_bits candidates: scan ascending windows of 2 * 2^pow_bits (86 % hit rate each) so that little work
    // is wasted; the window's atomicMin keeps the result the global minimum
    // (a window is at most 2^30 candidates: the grid dimension is 32 bits)
    // With several proofs in flight the extra round trips of a smaller window are hidden and the candidates hashed beyond the
    // witness are what costs: windows of 2^pow_bits (104 k candidates expected in 1.6 launches instead of 152 k in 1.2; measured
    // with 16 in flight over alternating 1920-proof runs: 294.5 / 296.6 / 296.0 / 294.9 proofs/s for windows of 2, 1, 1/2, 1/4 x 2^bits).
    static const int pow_window_env = [] { const char* e = getenv("GL_POW_WINDOW_LOG"); return e ? atoi(e) : 99; }();      // tuning knob: window = 2^(bits + this)
    const int wlog = pow_window_env != 99 ? pow_window_env : (gl_proofs_in_flight.load(std::memory_order_relaxed) > 2 ? 0 : 1);
    const int wbits = (int)min_leading_zeros + wlog < 8 ? 8 : (int)min_leading_zeros + wlog;
    const uint64_t batch = wbits >= 30 ? (uint64_t(1) << 30) : (uint64_t(1) << wbits);
    unsigned long long res = ~0ull;
    ctx->timing_begin("find proof-of-work witness");
    for (uint64_t base = 0; base < GL_P; base += batch) {
        GL_CHECK_HIP(hipMemsetAsync(d_res, 0xFF, sizeof(unsigned long long), st));
        pw.base = base; pw.count = (GL_P - base < batch) ? GL_P - base : batch;
        hipLaunchKernelGGL(k_pow_grind, dim3((unsigned)((pw.count + 255) / 256)), dim3(256), 0, st, pw);
        GL_CHECK_HIP(hipGetLastError());
        GL_TRY(d2h(ctx, &res, d_res, sizeof res));
        if (res != ~0ull) break;
    }
    ctx->timing_end();
    GL_REQUIRE(res != ~0ull, GL_ERR_INTERNAL, "Proof of work failed. This is highly unlikely!");
    *witness = (uint64_t)res;
    return GL_OK;
}

// ---- fri_prover_query_rounds (fri/prover.rs:162-216): the serialised FriQueryRound list
//      (util/serialization/mod.rs:1477-1546: per query 4 x (leaf, u8 path length, siblings), then per reduction the
//      `arity` extension evaluations and the path) ----
static int fri_query_blob(gl_fri* f, const uint32_t* x_index, uint32_t nq, std::vector<uint8_t>& o) {
    gl_ctx* ctx = f->ctx;
    const gl_circuit_desc& d = f->desc;
    hipStream_t st = ctx->stream;
    const uint32_t lgN = f->lgN;
    const size_t N = size_t(1) << lgN;
    GL_REQUIRE(f->round == d.num_fri_rounds && !f->committed, GL_ERR_ARG, "gl_fri_query: reduction rounds not finished");
    for (uint32_t q = 0; q < nq; q++) GL_REQUIRE(x_index[q] < N, GL_ERR_ARG, "gl_fri_query: query index out of range");
    // staging layout (u64 words): per oracle: rows [nq][ncols], paths [nq][levels][4]; per FRI round: leaves [nq][arity][2], paths
    struct Piece { size_t off, words; };
    std::vector<Piece> row_piece(4), path_piece(4), fleaf_piece(d.num_fri_rounds), fpath_piece(d.num_fri_rounds);
    size_t total = 0;
    const uint32_t init_levels = lgN - d.cap_height;
    for (int o2 = 0; o2 < 4; o2++) {
        row_piece[o2] = {total, nq * f->oracles[o2]->ncols}; total += row_piece[o2].words;
        path_piece[o2] = {total, (size_t)nq * init_levels * 4}; total += path_piece[o2].words;
    }
    for (unsigned r = 0; r < d.num_fri_rounds; r++) {
        const uint32_t ab = d.fri_arity_bits[r], lv = f->lg[r] - ab - d.cap_height;
        fleaf_piece[r] = {total, (size_t)nq * (2u << ab)}; total += fleaf_piece[r].words;
        fpath_piece[r] = {total, (size_t)nq * lv * 4}; total += fpath_piece[r].words;
    }
    DevBuf d_stage(ctx), d_idx(ctx);
    GL_TRY(d_stage.alloc((total + 8) * sizeof(gl_t)));
    GL_TRY(d_idx.alloc((size_t)nq * (2 + d.num_fri_rounds) * sizeof(uint32_t)));
    std::vector<uint32_t> idx_host((size_t)nq * (2 + d.num_fri_rounds));
    for (uint32_t q = 0; q < nq; q++) { idx_host[q] = x_index[q]; idx_host[nq + q] = host_bitrev32(x_index[q], lgN); }
    {
        std::vector<uint32_t> xi(x_index, x_index + nq);
        for (unsigned r = 0; r < d.num_fri_rounds; r++) for (uint32_t q = 0; q < nq; q++) { xi[q] >>= d.fri_arity_bits[r]; idx_host[(size_t)(2 + r) * nq + q] = xi[q]; }
    }
    GL_TRY(h2d_async(ctx, d_idx.p, idx_host.data(), idx_host.size() * sizeof(uint32_t)));
    const uint32_t* d_leaf = d_idx.as<uint32_t>();            // Merkle leaf indices
    const uint32_t* d_rows = d_leaf + nq;                     // natural LDE rows = bitrev(leaf)
    gl_t* stage = d_stage.as<gl_t>();
    ctx->timing_begin("FRI query gathers");
    for (int o2 = 0; o2 < 4; o2++) {
        const gl_batch* b = f->oracles[o2];
        const uint64_t* d_lo = nullptr;
        GL_TRY(ctx->get_offsets_table(b->tree.level_off.data(), b->tree.level_off.size(), &d_lo));
        unsigned cnt = nq * (unsigned)b->ncols;
        hipLaunchKernelGGL(k_gather_rows, dim3((cnt + 255) / 256), dim3(256), 0, st, b->lde, (uint64_t)N, (uint32_t)b->ncols, d_rows, nq, stage + row_piece[o2].off);
        cnt = nq * init_levels * 4;
        if (cnt) hipLaunchKernelGGL(k_gather_paths, dim3((cnt + 255) / 256), dim3(256), 0, st, b->tree.digests, d_lo, init_levels, d_leaf, nq, stage + path_piece[o2].off);
    }
    for (unsigned r = 0; r < d.num_fri_rounds; r++) {
        const uint32_t ab = d.fri_arity_bits[r], lv = f->lg[r] - ab - d.cap_height;
        const GlMerkle& t = f->trees[r]->m;
        const uint64_t* d_lo = nullptr;
        GL_TRY(ctx->get_offsets_table(t.level_off.data(), t.level_off.size(), &d_lo));
        const gl_t* va = f->vals[r]->as<gl_t>(); const gl_t* vb = va + (size_t(1) << f->lg[r]);
        const uint32_t* d_fl = d_idx.as<uint32_t>() + (size_t)(2 + r) * nq;
        unsigned cnt = nq << ab;
        hipLaunchKernelGGL(k_gather_fri_leaves, dim3((cnt + 255) / 256), dim3(256), 0, st, va, vb, f->lg[r], ab, d_fl, nq, stage + fleaf_piece[r].off);
        cnt = nq * lv * 4;
        if (cnt) hipLaunchKernelGGL(k_gather_paths, dim3((cnt + 255) / 256), dim3(256), 0, st, t.digests, d_lo, lv, d_fl, nq, stage + fpath_piece[r].off);
    }
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());
    std::vector<gl_t> host_stage(total + 8);
    GL_TRY(d2h(ctx, host_stage.data(), stage, total * sizeof(gl_t)));      // also orders the idx_host upload before it dies
    for (uint32_t q = 0; q < nq; q++) {
        for (int oi = 0; oi < 4; oi++) {
            const size_t nc = f->oracles[oi]->ncols;
            put_words(o, host_stage.data() + row_piece[oi].off + (size_t)q * nc, nc);
            o.push_back((uint8_t)init_levels);
            put_words(o, host_stage.data() + path_piece[oi].off + (size_t)q * init_levels * 4, (size_t)init_levels * 4);
        }
        for (unsigned r = 0; r < d.num_fri_rounds; r++) {
            const uint32_t ab = d.fri_arity_bits[r], lv = f->lg[r] - ab - d.cap_height;
            put_words(o, host_stage.data() + fleaf_piece[r].off + (size_t)q * (2u << ab), 2u << ab);
            o.push_back((uint8_t)lv);
            put_words(o, host_stage.data() + fpath_piece[r].off + (size_t)q * lv * 4, (size_t)lv * 4);
        }
    }
    return GL_OK;
}
extern "C" int gl_fri_query(gl_fri* f, const uint32_t* x_index, uint32_t num_queries, uint8_t* h_blob, size_t cap_bytes, size_t* num_bytes) {
    GL_REQUIRE(f && x_index && num_bytes, GL_ERR_ARG, "gl_fri_query: null argument");
    GL_REQUIRE(num_queries >= 1 && num_queries <= 256, GL_ERR_ARG, "gl_fri_query: 1..256 queries");
    GL_TRY(f->ctx->activate());
    std::vector<uint8_t> blob;
    GL_TRY(fri_query_blob(f, x_index, num_queries, blob));
    *num_bytes = blob.size();
    if (!h_blob) return GL_OK;
    GL_REQUIRE(cap_bytes >= blob.size(), GL_ERR_ARG, "gl_fri_query: output too small");
    memcpy(h_blob, blob.data(), blob.size());
    return GL_OK;
}

// ======================================================================================================================
// prove(): the driver (plonk/prover.rs:102-329) -- the phases above plus the transcript
// ======================================================================================================================
static int prove_impl(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* h_wires, bool wires_on_device, const uint64_t* h_pis, size_t npis, const uint64_t* h_pi_hash, gl_proof** out);
extern "C" int gl_prove(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* h_wires, const uint64_t* h_pis, size_t npis, gl_proof** out) {
    return prove_impl(ctx, cir, h_wires, false, h_pis, npis, nullptr, out);
}
// the witness as the reference holds it: one host vector per wire (MatrixWitness.wire_values: Vec<Vec<F>>, iop/witness.rs:256-258)
extern "C" int gl_prove_columns(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* const* h_wire_columns, const uint64_t* h_pis, size_t npis, gl_proof** out) {
    GL_REQUIRE(ctx && cir && h_wire_columns && out, GL_ERR_ARG, "gl_prove_columns: null argument");
    GL_TRY(ctx->activate());
    const size_t n = cir->n, nw = cir->desc.num_wires;
    for (size_t c = 0; c < nw; c++) GL_REQUIRE(h_wire_columns[c], GL_ERR_ARG, "gl_prove_columns: null column");
    DevBuf d_wit(ctx); GL_TRY(d_wit.alloc(nw * n * sizeof(gl_t)));
    ctx->timing_begin("H2D witness");
    GL_TRY(h2d_witness(ctx, d_wit.as<gl_t>(), nw, n, h_wire_columns, nullptr));
    ctx->timing_end();
    // the caller's vectors have been read when h2d_witness returns; the proof is finished before d_wit is released
    return prove_impl(ctx, cir, d_wit.as<uint64_t>(), true, h_pis, npis, nullptr, out);
}
extern "C" int gl_prove_device(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* d_wires, const uint64_t* h_pis, size_t npis, gl_proof** out) {
    return prove_impl(ctx, cir, d_wires, true, h_pis, npis, nullptr, out);
}
extern "C" int gl_prove_device_hashed(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* d_wires, const uint64_t* h_pis, size_t npis,
                                      const uint64_t public_inputs_hash[4], gl_proof** out) {
    GL_REQUIRE(public_inputs_hash, GL_ERR_ARG, "gl_prove_device_hashed: null hash");
    return prove_impl(ctx, cir, d_wires, true, h_pis, npis, public_inputs_hash, out);
}
// One pass of the proving pipeline over an all-zero witness, result thrown away: afterwards this context holds everything a proof of this
// circuit needs besides its own data -- the code objects of every kernel on the path loaded, the twiddle / power tables of the circuit's
// transform sizes built, the context's pool grown to the pipeline's working set.  (The reference precomputes its fft_root_table in build()
// too, circuit_builder.rs:1016-1019.)  A zero witness does not satisfy the circuit; nothing on the path asserts that it does.
extern "C" int gl_circuit_warm_up(gl_ctx* ctx, const gl_circuit* cir) {
    GL_REQUIRE(ctx && cir, GL_ERR_ARG, "gl_circuit_warm_up: null argument");
    GL_REQUIRE(cir->ctx->device == ctx->device, GL_ERR_ARG, "gl_circuit_warm_up: circuit lives on another device");
    GL_TRY(ctx->activate());
    const size_t n = cir->n;
    DevBuf d_w(ctx); GL_TRY(d_w.alloc(135 * n * sizeof(gl_t)));
    GL_CHECK_HIP(hipMemsetAsync(d_w.p, 0, 135 * n * sizeof(gl_t), ctx->stream));
    std::vector<uint64_t> pis(cir->desc.num_public_inputs ? cir->desc.num_public_inputs : 1, 0);
    gl_proof* pr = nullptr;
    const int st = prove_impl(ctx, cir, d_w.as<uint64_t>(), true, pis.data(), cir->desc.num_public_inputs, nullptr, &pr);
    if (pr) gl_proof_free(pr);
    return st == GL_ERR_ZETA_IN_SUBGROUP ? GL_OK : st;       // (probability 2^-49: still warmed up to the opening point)
}
static int prove_impl(gl_ctx* ctx, const gl_circuit* cir, const uint64_t* h_wires, bool wires_on_device, const uint64_t* h_pis, size_t npis, const uint64_t* h_pi_hash, gl_proof** out) {
    GL_REQUIRE(ctx && cir && h_wires && h_pis && out, GL_ERR_ARG, "gl_prove: null argument");
    // circuit data is read-only while proving: any context (stream) of the same device may prove against it
    GL_REQUIRE(cir->ctx->device == ctx->device, GL_ERR_ARG, "gl_prove: circuit lives on another device");
    const gl_circuit_desc& d = cir->desc;
    GL_REQUIRE(npis == d.num_public_inputs, GL_ERR_ARG, "gl_prove: wrong number of public inputs");
    GL_TRY(ctx->activate());
    struct InFlight { InFlight() { gl_proofs_in_flight.fetch_add(1, std::memory_order_relaxed); } ~InFlight() { gl_proofs_in_flight.fetch_sub(1, std::memory_order_relaxed); } } in_flight;
    const size_t n = cir->n, N = n << d.rate_bits;
    const uint32_t lgn = d.degree_bits, ncap = 4u << d.cap_height;
    hipStream_t st = ctx->stream;
    std::unique_ptr<gl_proof> proof(new gl_proof());
    std::vector<gl_t> h_apow_quot;          // sources of async uploads: alive until the function returns (after the last sync)
    std::vector<const gl_t*> h_open_cols;

    // ---- 4. wires commitment (prover.rs:145-156) ----
    DevBuf d_wit(ctx);
    const gl_t* d_wires = (const gl_t*)h_wires;
    if (!wires_on_device) {
        GL_TRY(d_wit.alloc(135 * n * sizeof(gl_t)));
        ctx->timing_begin("H2D witness");
        GL_TRY(h2d_witness(ctx, d_wit.as<gl_t>(), 135, n, nullptr, h_wires));
        ctx->timing_end();
        d_wires = d_wit.as<gl_t>();
    }
    BatchHolder wires; GL_TRY(gl_batch_from_device(ctx, d_wires, 135, n, d.rate_bits, d.cap_height, 1, &wires.b));
    // public_inputs_hash (prover.rs:126-127) on the host while the GPU commits
    gl_t pi_hash[4];
    if (h_pi_hash) for (int i = 0; i < 4; i++) pi_hash[i] = gl_canon(h_pi_hash[i]);      // the witness generator's sponge already produced it
    else glhost::host_hash_no_pad(h_pis, npis, pi_hash);
    HostChallenger ch;
    ch.observe_many(cir->circuit_digest, 4);
    ch.observe_many(pi_hash, 4);
    std::vector<gl_t> cap(ncap);
    GL_TRY(gl_batch_cap(wires.b, cap.data()));
    proof->caps.insert(proof->caps.end(), cap.begin(), cap.end());
    ch.observe_many(cap.data(), ncap);
    gl_t betas[2], gammas[2], alphas[2];
    for (int i = 0; i < 2; i++) betas[i] = ch.challenge();
    for (int i = 0; i < 2; i++) gammas[i] = ch.challenge();
    // lookups: four coins per challenge, [betas | gammas | 4 more] (prover.rs:166-184)
    const size_t nlk = 2 * (size_t)d.num_lookup_polys, nzs = 20 + nlk;
    gl_t deltas[8] = {betas[0], betas[1], gammas[0], gammas[1], 0, 0, 0, 0};
    if (nlk) for (int i = 4; i < 8; i++) deltas[i] = ch.challenge();

    // ---- 6/7. partial products and Z (and the lookup polynomials), commitment (prover.rs:189-223) ----
    BatchHolder zs;
    {
        DevBuf d_zs(ctx); GL_TRY(d_zs.alloc(nzs * n * sizeof(gl_t)));
        GL_TRY(partial_products_values(ctx, cir, d_wires, betas, gammas, d_zs.as<gl_t>()));
        if (nlk) GL_TRY(lookup_polys_values(ctx, cir, d_wires, deltas, d_zs.as<gl_t>() + 20 * n));
        if (ctx->capture_intermediates) { proof->zs_pp.resize(nzs * n); GL_TRY(d2h(ctx, proof->zs_pp.data(), d_zs.p, nzs * n * sizeof(gl_t))); }
        GL_TRY(gl_batch_from_device(ctx, d_zs.as<uint64_t>(), nzs, n, d.rate_bits, d.cap_height, 1, &zs.b));
    }
    d_wit.release();                                                            // stream-ordered: the kernels above are already queued
    GL_TRY(gl_batch_cap(zs.b, cap.data()));
    proof->caps.insert(proof->caps.end(), cap.begin(), cap.end());
    ch.observe_many(cap.data(), ncap);
    for (int i = 0; i < 2; i++) alphas[i] = ch.challenge();

    // ---- 9/10. quotient polynomials (prover.rs:229-271) ----
    BatchHolder quot;
    {
        DevBuf d_q(ctx); GL_TRY(d_q.alloc(2 * N * sizeof(gl_t)));
        GL_TRY(quotient_chunks(ctx, cir, wires.b, zs.b, pi_hash, betas, gammas, alphas, d_q.as<gl_t>(), h_apow_quot, nlk ? deltas : nullptr));
        if (ctx->capture_intermediates) { proof->quotient.resize(16 * n); GL_TRY(d2h(ctx, proof->quotient.data(), d_q.p, 16 * n * sizeof(gl_t))); }
        GL_TRY(gl_batch_from_device(ctx, d_q.as<uint64_t>(), 16, n, d.rate_bits, d.cap_height, 0, &quot.b));
    }
    GL_TRY(gl_batch_cap(quot.b, cap.data()));
    proof->caps.insert(proof->caps.end(), cap.begin(), cap.end());
    ch.observe_many(cap.data(), ncap);

    // ---- 11. zeta (prover.rs:273-283) ----
    gl2_t zeta; zeta.a = ch.challenge(); zeta.b = ch.challenge();
    {
        gl2_t zn = zeta;
        for (uint32_t i = 0; i < lgn; i++) zn = gl2_mul(zn, zn);
        zn = gl2_canon(zn);
        if (zn.a == 1 && zn.b == 0) return gl_fail(GL_ERR_ZETA_IN_SUBGROUP, "Opening point is in the subgroup.", __FILE__, __LINE__);
    }
    const gl2_t gzeta = gl2_canon(gl2_scalar(zeta, gl_host_root_of_unity(lgn)));

    // ---- 12. openings (proof.rs:306-344): all five evaluations behind one sync ----
    const gl_batch* oracles[4] = {cir->cs_batch, wires.b, zs.b, quot.b};
    const size_t ncs = d.num_constants + 80, nopen = ncs + 135 + 20 + 16 + nlk, nnext = 2 + nlk;
    std::vector<gl_t> open_zeta(2 * nopen), open_next(2 * nnext);
    {
        GL_TRY(ctx->ensure_dev_small(1 << 20));
        gl_t* d_open = ctx->dev_small + 4096;
        ctx->timing_begin("construct the opening set");
        // zeta^i and (g zeta)^i tabulated once per proof, shared by all polynomials
        DevBuf d_zpow(ctx); GL_TRY(d_zpow.alloc(4 * n * sizeof(gl_t)));
        gl_t* zp = d_zpow.as<gl_t>();
        hipLaunchKernelGGL(k_ext_powers2, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, st, zeta.a, zeta.b, gzeta.a, gzeta.b, (uint32_t)n, zp);
        // one launch for all 257 openings: the list of coefficient columns goes up as a small pointer table
        // FriOpenings order (proof.rs:346-380): constants, sigmas, wires, zs, partial products, quotient, lookups | zs_next, lookups_next
        for (int o = 0; o < 4; o++) for (size_t c = 0; c < (o == 2 ? (size_t)20 : oracles[o]->ncols); c++) h_open_cols.push_back(oracles[o]->coeffs + c * n);
        for (size_t c = 0; c < nlk; c++) h_open_cols.push_back(zs.b->coeffs + (20 + c) * n);
        h_open_cols.push_back(zs.b->coeffs); h_open_cols.push_back(zs.b->coeffs + n);
        for (size_t c = 0; c < nlk; c++) h_open_cols.push_back(zs.b->coeffs + (20 + c) * n);
        DevBuf d_open_cols(ctx); GL_TRY(d_open_cols.alloc(h_open_cols.size() * sizeof(gl_t*)));
        GL_TRY(h2d_async(ctx, d_open_cols.p, h_open_cols.data(), h_open_cols.size() * sizeof(gl_t*)));
        hipLaunchKernelGGL(k_eval_list_with_powers, dim3((unsigned)h_open_cols.size()), dim3(256), 0, st, d_open_cols.as<const gl_t*>(), (uint32_t)n,
                           (uint32_t)nopen, zp, zp + n, zp + 2 * n, zp + 3 * n, d_open);
        ctx->timing_end();
        GL_CHECK_HIP(hipGetLastError());
        std::vector<gl_t> tmp(2 * (nopen + nnext));
        GL_TRY(d2h(ctx, tmp.data(), d_open, tmp.size() * sizeof(gl_t)));
        memcpy(open_zeta.data(), tmp.data(), 2 * nopen * sizeof(gl_t));
        memcpy(open_next.data(), tmp.data() + 2 * nopen, 2 * nnext * sizeof(gl_t));
    }
    // FriOpenings order = oracle order: constants, sigmas, wires, zs, partial products, quotient; then zs_next
    ch.observe_many(open_zeta.data(), open_zeta.size());
    ch.observe_many(open_next.data(), open_next.size());

    // ---- 14. prove_openings (fri/oracle.rs:162-219), fri_proof (fri/prover.rs:20-66) ----
    gl_t fri_alpha[2], zeta_w[2] = {zeta.a, zeta.b};
    fri_alpha[0] = ch.challenge(); fri_alpha[1] = ch.challenge();
    gl_fri* fri_raw = nullptr;
    GL_TRY(gl_fri_combine(ctx, cir, oracles, zeta_w, fri_alpha, &fri_raw));
    std::unique_ptr<gl_fri, void (*)(gl_fri*)> fri(fri_raw, gl_fri_free);
    std::vector<gl_t> fri_caps, fri_betas;
    for (unsigned r = 0; r < d.num_fri_rounds; r++) {
        GL_TRY(gl_fri_commit_round(fri.get(), cap.data()));
        fri_caps.insert(fri_caps.end(), cap.begin(), cap.end());
        ch.observe_many(cap.data(), ncap);
        gl_t beta[2]; beta[0] = ch.challenge(); beta[1] = ch.challenge();
        fri_betas.push_back(beta[0]); fri_betas.push_back(beta[1]);
        GL_TRY(gl_fri_fold(fri.get(), beta));
    }
    size_t fin_words = 0;
    GL_TRY(gl_fri_final_poly(fri.get(), nullptr, 0, &fin_words));
    std::vector<gl_t> fin_il(fin_words);
    GL_TRY(gl_fri_final_poly(fri.get(), fin_il.data(), fin_il.size(), &fin_words));
    ch.observe_many(fin_il.data(), fin_il.size());
    gl_t pow_witness = 0;
    GL_TRY(gl_pow_grind(ctx, ch.state, ch.in, (uint32_t)ch.nin, d.proof_of_work_bits, &pow_witness));
    ch.observe(pow_witness);
    const gl_t pow_response = ch.challenge();
    GL_REQUIRE(pow_response == 0 || (uint32_t)__builtin_clzll(pow_response) >= d.proof_of_work_bits, GL_ERR_INTERNAL, "PoW response mismatch");
    const uint32_t nq = d.num_query_rounds;
    std::vector<uint32_t> x_index(nq);
    for (uint32_t q = 0; q < nq; q++) { x_index[q] = (uint32_t)(ch.challenge() % (uint64_t)N); proof->query_indices.push_back(x_index[q]); }
    std::vector<uint8_t> query_blob;
    query_blob.reserve(200000);
    GL_TRY(fri_query_blob(fri.get(), x_index.data(), nq, query_blob));

    // ---- 15. assemble ProofWithPublicInputs bytes (util/serialization/mod.rs:1939-1981) ----
    std::vector<uint8_t>& o = proof->bytes;
    o.reserve(query_blob.size() + 8 * (npis + 2 * nopen + fin_words + 4 * ncap) + 4096);
    put_words(o, proof->caps.data(), proof->caps.size());                        // wires_cap, zs_pp_cap, quotient_cap
    // OpeningSet (:1409-1423): constants, sigmas, wires, zs, zs_next, lookup_zs, lookup_zs_next, partial products, quotient
    {
        const gl_t* z = open_zeta.data();
        size_t o_cs = 0, o_w = 2 * ncs, o_z = o_w + 2 * 135, o_pp = o_z + 2 * 2, o_q = o_z + 2 * 20, o_lk = o_q + 2 * 16;
        put_words(o, z + o_cs, 2 * ncs);
        put_words(o, z + o_w, 2 * 135);
        put_words(o, z + o_z, 2 * 2);
        put_words(o, open_next.data(), 2 * 2);
        put_words(o, z + o_lk, 2 * nlk);
        put_words(o, open_next.data() + 2 * 2, 2 * nlk);
        put_words(o, z + o_pp, 2 * 18);
        put_words(o, z + o_q, 2 * 16);
    }
    put_words(o, fri_caps.data(), fri_caps.size());
    o.insert(o.end(), query_blob.begin(), query_blob.end());
    put_words(o, fin_il.data(), fin_il.size());
    put_u64(o, pow_witness);
    put_u64(o, npis);
    put_words(o, h_pis, npis);

    std::vector<gl_t>& cv = proof->challenges;
    for (int i = 0; i < 2; i++) cv.push_back(betas[i]);
    for (int i = 0; i < 2; i++) cv.push_back(gammas[i]);
    for (int i = 0; i < 2; i++) cv.push_back(alphas[i]);
    cv.push_back(zeta.a); cv.push_back(zeta.b); cv.push_back(fri_alpha[0]); cv.push_back(fri_alpha[1]); cv.push_back(pow_witness);
    for (int i = 0; i < 4; i++) cv.push_back(pi_hash[i]);
    cv.insert(cv.end(), fri_betas.begin(), fri_betas.end());
    GL_CHECK_HIP(gl_stream_wait(st));
    *out = proof.release();
    return GL_OK;
}

extern "C" size_t gl_proof_num_bytes(const gl_proof* p) { return p ? p->bytes.size() : 0; }
extern "C" int gl_proof_bytes(const gl_proof* p, uint8_t* h_out, size_t cap) {
    GL_REQUIRE(p && h_out && cap >= p->bytes.size(), GL_ERR_ARG, "buffer too small");
    memcpy(h_out, p->bytes.data(), p->bytes.size());
    return GL_OK;
}
extern "C" size_t gl_proof_challenges(const gl_proof* p, uint64_t* h_out) {
    if (!p || !h_out) return 0;
    memcpy(h_out, p->challenges.data(), p->challenges.size() * sizeof(gl_t));
    return p->challenges.size();
}
extern "C" int gl_proof_caps(const gl_proof* p, uint64_t* h_out) {
    GL_REQUIRE(p && h_out, GL_ERR_ARG, "null argument");
    memcpy(h_out, p->caps.data(), p->caps.size() * sizeof(gl_t));
    return GL_OK;
}
extern "C" int gl_ctx_capture_intermediates(gl_ctx* ctx, int enable) {
    GL_REQUIRE(ctx, GL_ERR_ARG, "null context");
    ctx->capture_intermediates = enable != 0;
    return GL_OK;
}
extern "C" int gl_proof_zs_partial_products(const gl_proof* p, uint64_t* h_out) {
    GL_REQUIRE(p && h_out, GL_ERR_ARG, "null argument");
    GL_REQUIRE(!p->zs_pp.empty(), GL_ERR_ARG, "intermediates were not captured: call gl_ctx_capture_intermediates(ctx, 1) before proving");
    memcpy(h_out, p->zs_pp.data(), p->zs_pp.size() * sizeof(gl_t));
    return GL_OK;
}
extern "C" int gl_proof_quotient_chunks(const gl_proof* p, uint64_t* h_out) {
    GL_REQUIRE(p && h_out, GL_ERR_ARG, "null argument");
    GL_REQUIRE(!p->quotient.empty(), GL_ERR_ARG, "intermediates were not captured: call gl_ctx_capture_intermediates(ctx, 1) before proving");
    memcpy(h_out, p->quotient.data(), p->quotient.size() * sizeof(gl_t));
    return GL_OK;
}
extern "C" size_t gl_proof_query_indices(const gl_proof* p, uint64_t* h_out) {
    if (!p || !h_out) return 0;
    memcpy(h_out, p->query_indices.data(), p->query_indices.size() * sizeof(uint64_t));
    return p->query_indices.size();
}
extern "C" void gl_proof_free(gl_proof* p) { delete p; }
