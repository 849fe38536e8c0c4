// Witness generation for the matmul circuit family with the wire matrix produced directly in HBM
// (the reference's generate_partial_witness + full_witness, plonky2/src/plonk/prover.rs:118-133, iop/generator.rs:19-98,
// specialised to the generators this circuit installs: ArithmeticBaseGenerator gates/arithmetic_base.rs:184-225,
// PoseidonGenerator gates/poseidon.rs:430-497, the PublicInputGate / ConstantGate copies, RandomValueGenerator
// iop/generator.rs and circuit_builder.rs:904-910).
//
// Split by what each side is good at:
//   * the m^3 ArithmeticGate operations are independent per output C[i][j]: one GPU lane per output walks its k-chain and
//     scatters the 4 wires of every multiply / add operation into the column-major matrix;
//   * the public-input hash is ONE sequential sponge of 3m^2/8 permutations whose every intermediate S-box input is a
//     wire: inherently serial, done on the host core that drives the context while the GPU fills the arithmetic rows,
//     then uploaded as one strided copy (the PoseidonGate rows, the PublicInputGate row and the ConstantGate row are
//     consecutive rows of the trace).
// The result is bit-identical to gl_matmul_witness (host) -- tests/test_gpu_parity.py.
#include "context.hpp"
#include "host_circuit.hpp"
#include <memory>

struct gl_matmul_witgen {
    gl_ctx* ctx = nullptr;
    const gl_host_circuit* hc = nullptr;                 // borrowed: must outlive the generator
    uint32_t* d_mul_row = nullptr; uint32_t* d_add_row = nullptr;
    gl_t* d_ab = nullptr;                                // a then b, m*m each
    gl_t* h_special = nullptr;                           // pinned [135][R] staging of the non-arithmetic rows
    gl_t* h_ab = nullptr;                                // pinned copy of the operands
    size_t R = 0;
    std::vector<gl_t> cvals;
};

// one lane per output (i, j): the reference's loop order `for i, for j, for k` (matrix_mul.rs:46-58) fixes the operation
// indices: multiply t = (i*m + j)*m + k, add t' = (i*m + j)*(m-1) + (k-1); operation t sits in slot t % 20 of row-block
// t / 20 of its kind (gadgets/arithmetic.rs:87-99)
__global__ void k_matmul_arith_rows(const gl_t* __restrict__ a, const gl_t* __restrict__ b, uint32_t m, uint64_t n,
                                    const uint32_t* __restrict__ mul_row, const uint32_t* __restrict__ add_row, gl_t* __restrict__ wires) {
    const uint32_t ij = blockIdx.x * blockDim.x + threadIdx.x;
    if (ij >= m * m) return;
    const uint32_t i = ij / m, j = ij - i * m;
    uint64_t mc = (uint64_t)ij * m, ac = (uint64_t)ij * (m - 1);
    gl_t cur = 0;
    for (uint32_t k = 0; k < m; k++, mc++) {
        const gl_t x = gl_canon(a[i * m + k]), y = gl_canon(b[k * m + j]), p = gl_canon(gl_mul(x, y));
        {   // mul(x, y) = arithmetic(1, 0, x, y, x): wires multiplicand_0, multiplicand_1, addend, output
            const uint64_t row = mul_row[mc / 20]; const uint32_t s = (uint32_t)(mc % 20);
            gl_t* w = wires + (uint64_t)(4 * s) * n + row;
            w[0] = x; w[n] = y; w[2 * n] = x; w[3 * n] = p;
        }
        if (k == 0) { cur = p; continue; }
        const gl_t sum = gl_canon(gl_add(cur, p));
        {   // add(cur, p) = arithmetic(1, 1, cur, one, p)
            const uint64_t row = add_row[ac / 20]; const uint32_t s = (uint32_t)(ac % 20);
            gl_t* w = wires + (uint64_t)(4 * s) * n + row;
            w[0] = cur; w[n] = 1; w[2 * n] = p; w[3 * n] = sum;
        }
        ac++;
        cur = sum;
    }
}

extern "C" void gl_matmul_witgen_free(gl_matmul_witgen* g) {
    if (!g) return;
    if (g->ctx) {
        (void)g->ctx->activate();
        (void)gl_stream_wait(g->ctx->stream);
        if (g->d_mul_row) g->ctx->pool_release(g->d_mul_row);
        if (g->d_add_row) g->ctx->pool_release(g->d_add_row);
        if (g->d_ab) g->ctx->pool_release(g->d_ab);
    }
    if (g->h_special) (void)hipHostFree(g->h_special);
    if (g->h_ab) (void)hipHostFree(g->h_ab);
    gl_ctx_release(g->ctx);
    delete g;
}

extern "C" int gl_matmul_witgen_create(gl_ctx* ctx, const gl_host_circuit* hc, gl_matmul_witgen** out) {
    GL_REQUIRE(ctx && hc && out, GL_ERR_ARG, "gl_matmul_witgen_create: null argument");
    GL_TRY(ctx->activate());
    const glhost::HostCircuit& h = hc->hc;
    GL_REQUIRE(h.constant_row == h.first_poseidon_row + h.num_poseidon_rows + 1 && h.pi_row + 1 == h.constant_row, GL_ERR_INTERNAL,
               "matmul trace layout: PoseidonGate rows, PublicInputGate row and ConstantGate row must be consecutive");
    std::unique_ptr<gl_matmul_witgen, void (*)(gl_matmul_witgen*)> g(new gl_matmul_witgen(), gl_matmul_witgen_free);
    g->ctx = ctx; ctx->retain(); g->hc = hc; g->R = h.num_poseidon_rows + 2;
    const size_t mm = h.m * h.m;
    GL_TRY(ctx->pool_alloc((h.mul_row.size() + 1) * sizeof(uint32_t), (void**)&g->d_mul_row));
    GL_TRY(ctx->pool_alloc((h.add_row.size() + 1) * sizeof(uint32_t), (void**)&g->d_add_row));
    GL_TRY(ctx->pool_alloc(2 * mm * sizeof(gl_t), (void**)&g->d_ab));
    GL_CHECK_HIP(hipHostMalloc((void**)&g->h_special, 135 * g->R * sizeof(gl_t), hipHostMallocDefault));
    GL_CHECK_HIP(hipHostMalloc((void**)&g->h_ab, 2 * mm * sizeof(gl_t), hipHostMallocDefault));
    GL_TRY(gl_copy_h2d(ctx, g->d_mul_row, h.mul_row.data(), h.mul_row.size() * sizeof(uint32_t)));
    if (!h.add_row.empty()) GL_TRY(gl_copy_h2d(ctx, g->d_add_row, h.add_row.data(), h.add_row.size() * sizeof(uint32_t)));
    g->cvals.resize(mm);
    *out = g.release();
    return GL_OK;
}

extern "C" int gl_matmul_witgen_run(gl_matmul_witgen* g, const uint64_t* a, const uint64_t* b, uint64_t filler_seed, uint64_t* d_wires, uint64_t* h_pis,
                                    uint64_t* h_pi_hash) {
    GL_REQUIRE(g && a && b && d_wires && h_pis, GL_ERR_ARG, "gl_matmul_witgen_run: null argument");
    gl_ctx* ctx = g->ctx;
    GL_TRY(ctx->activate());
    const glhost::HostCircuit& h = g->hc->hc;
    const size_t m = h.m, n = h.n, mm = m * m, R = g->R;
    hipStream_t st = ctx->stream;
    using namespace glhost;

    // --- GPU: zero the trace, then the arithmetic rows ---
    for (size_t t = 0; t < mm; t++) { g->h_ab[t] = gl_canon(a[t]); g->h_ab[mm + t] = gl_canon(b[t]); }
    ctx->timing_begin("witness: arithmetic rows");
    GL_CHECK_HIP(hipMemsetAsync(d_wires, 0, 135 * n * sizeof(gl_t), st));
    GL_CHECK_HIP(hipMemcpyAsync(g->d_ab, g->h_ab, 2 * mm * sizeof(gl_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_matmul_arith_rows, dim3((unsigned)((mm + 63) / 64)), dim3(64), 0, st, g->d_ab, g->d_ab + mm, (uint32_t)m, (uint64_t)n,
                       g->d_mul_row, g->d_add_row, (gl_t*)d_wires);
    ctx->timing_end();
    GL_CHECK_HIP(hipGetLastError());

    // --- host, meanwhile: C, the public inputs, the PI-hash sponge rows, the PublicInputGate and ConstantGate rows ---
    const gl_t* ha = g->h_ab; const gl_t* hb = g->h_ab + mm;
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < m; j++) {
            gl_t cur = 0;
            for (size_t k = 0; k < m; k++) {
                const gl_t p = gl_canon(gl_mul(ha[i * m + k], hb[k * m + j]));
                cur = k ? gl_canon(gl_add(cur, p)) : p;
            }
            g->cvals[i * m + j] = cur;
        }
    const size_t n_pi = 3 * mm;
    for (size_t ij = 0; ij < mm; ij++) { h_pis[3 * ij] = ha[ij]; h_pis[3 * ij + 1] = hb[ij]; h_pis[3 * ij + 2] = g->cvals[ij]; }
    gl_t* sp = g->h_special;                               // sp[col * R + r], r = row - first_poseidon_row
    memset(sp, 0, 135 * R * sizeof(gl_t));
    gl_t state[12] = {0};
    for (size_t pr = 0; pr < h.num_poseidon_rows; pr++) {
        const size_t off = pr * 8, c = std::min<size_t>(8, n_pi - off);
        for (size_t t = 0; t < c; t++) state[t] = h_pis[off + t];
        poseidon_row_witness(state, sp + pr, R);
        for (int t = 0; t < 12; t++) state[t] = sp[(PW_OUTPUT + t) * R + pr];
    }
    const size_t r_pi = h.num_poseidon_rows, r_const = r_pi + 1;
    for (int t = 0; t < 4; t++) sp[t * R + r_pi] = state[t];
    if (h_pi_hash) for (int t = 0; t < 4; t++) h_pi_hash[t] = state[t];          // = hash_no_pad(public inputs) (prover.rs:126-127)
    { uint64_t s = filler_seed; for (size_t c = 4; c < 135; c++) sp[c * R + r_pi] = splitmix64_next(s) % GL_P; }   // circuit_builder.rs:904-910
    sp[0 * R + r_const] = 0; sp[1 * R + r_const] = 1;

    // --- one strided upload of the R consecutive special rows (after the memset on the same stream) ---
    ctx->timing_begin("witness: upload hash rows");
    GL_CHECK_HIP(hipMemcpy2DAsync((gl_t*)d_wires + h.first_poseidon_row, n * sizeof(gl_t), sp, R * sizeof(gl_t), R * sizeof(gl_t), 135,
                                  hipMemcpyHostToDevice, st));
    ctx->timing_end();
    GL_CHECK_HIP(gl_stream_wait(st));                // the pinned staging buffers are reused by the next call
    return GL_OK;
}

// ======================================================================================================================
// Prover pool: many independent proofs in flight on one GPU from ONE call -- the C++ counterpart of the reference fanning a
// batch out over its Rayon pool.  A pool owns one device-resident circuit and `lanes` contexts (HIP stream + allocator +
// witness generator each); gl_prover_pool_prove_matmul hands item i to lane i % lanes on `lanes` host threads, so that the
// transcript round trips and the host-side hash sponge of one proof hide behind the kernels of the others.
// ======================================================================================================================
#include <thread>
#include <atomic>

struct gl_prover_pool {
    int device = 0;
    const gl_host_circuit* hc = nullptr;                 // borrowed
    std::vector<gl_ctx*> ctxs;
    gl_circuit* circuit = nullptr;                       // owned by ctxs[0]
    std::vector<gl_matmul_witgen*> gens;
    std::vector<uint64_t*> d_wires;                      // one witness matrix per lane
};

extern "C" void gl_prover_pool_free(gl_prover_pool* p) {
    if (!p) return;
    for (size_t k = 0; k < p->ctxs.size(); k++) {
        if (k < p->gens.size() && p->gens[k]) gl_matmul_witgen_free(p->gens[k]);
        if (k < p->d_wires.size() && p->d_wires[k]) (void)gl_dev_free(p->ctxs[k], p->d_wires[k]);
    }
    if (p->circuit) gl_circuit_free(p->circuit);
    for (gl_ctx* c : p->ctxs) gl_ctx_destroy(c);
    delete p;
}

// The same pool for ANY circuit the library proves (a description + the constants || sigmas value columns, as gl_circuit_create takes
// them): no witness generators, the witnesses come from the host (gl_prover_pool_prove_columns).  Every lane is warmed up.
extern "C" int gl_prover_pool_create_generic(int device, const gl_circuit_desc* desc, const uint64_t* h_constants_sigmas, uint32_t lanes, gl_prover_pool** out) {
    GL_REQUIRE(desc && h_constants_sigmas && out && lanes >= 1 && lanes <= 64, GL_ERR_ARG, "gl_prover_pool_create_generic: bad argument (1..64 lanes)");
    std::unique_ptr<gl_prover_pool, void (*)(gl_prover_pool*)> p(new gl_prover_pool(), gl_prover_pool_free);
    p->device = device;
    for (uint32_t k = 0; k < lanes; k++) {
        gl_ctx* c = nullptr;
        GL_TRY(gl_ctx_create(device, nullptr, &c));
        p->ctxs.push_back(c);
    }
    GL_TRY(gl_circuit_create(p->ctxs[0], desc, h_constants_sigmas, &p->circuit));
    for (uint32_t k = 0; k < lanes; k++) GL_TRY(gl_circuit_warm_up(p->ctxs[k], p->circuit));
    *out = p.release();
    return GL_OK;
}

// count proofs from HOST witnesses: columns[i][j] = wire column j (n values) of witness i, as MatrixWitness.wire_values holds them
// (iop/witness.rs:256-258); public_inputs[i] = its num_public_inputs values.  Item i is proved on lane i % lanes through gl_prove_columns
// (the lane's pinned H2D ring).  out_proofs[i] receives a gl_proof; returns the first error of any lane (the other proofs stay valid).
extern "C" int gl_prover_pool_prove_columns(gl_prover_pool* p, size_t count, const uint64_t* const* const* columns, const uint64_t* const* public_inputs,
                                            gl_proof** out_proofs) {
    GL_REQUIRE(p && p->circuit && (count == 0 || (columns && public_inputs && out_proofs)), GL_ERR_ARG, "gl_prover_pool_prove_columns: null argument");
    for (size_t i = 0; i < count; i++) out_proofs[i] = nullptr;
    gl_circuit_desc d;
    GL_TRY(gl_circuit_description(p->circuit, &d));
    const size_t lanes = p->ctxs.size(), npis = d.num_public_inputs;
    std::atomic<int> first_error{GL_OK};
    std::vector<std::string> messages(lanes);
    const uint64_t none = 0;
    auto work = [&](size_t lane) {
        for (size_t i = lane; i < count && first_error.load() == GL_OK; i += lanes) {
            int st = (columns[i] && (public_inputs[i] || npis == 0)) ? GL_OK : GL_ERR_ARG;
            if (st == GL_OK) st = gl_prove_columns(p->ctxs[lane], p->circuit, columns[i], npis ? public_inputs[i] : &none, npis, &out_proofs[i]);
            if (st != GL_OK) { int expected = GL_OK; messages[lane] = st == GL_ERR_ARG && !columns[i] ? "gl_prover_pool_prove_columns: null witness" : gl_last_error(); first_error.compare_exchange_strong(expected, st); }
        }
    };
    std::vector<std::thread> threads;
    for (size_t k = 1; k < lanes && k < count; k++) threads.emplace_back(work, k);
    work(0);
    for (auto& t : threads) t.join();
    const int st = first_error.load();
    if (st != GL_OK) {
        for (auto& m : messages) if (!m.empty()) return gl_fail(st, m.c_str(), __FILE__, __LINE__);
        return gl_fail(st, "gl_prover_pool_prove_columns: a lane failed", __FILE__, __LINE__);
    }
    return GL_OK;
}

extern "C" int gl_prover_pool_create(int device, const gl_host_circuit* hc, uint32_t lanes, gl_prover_pool** out) {
    GL_REQUIRE(hc && out && lanes >= 1 && lanes <= 64, GL_ERR_ARG, "gl_prover_pool_create: bad argument (1..64 lanes)");
    std::unique_ptr<gl_prover_pool, void (*)(gl_prover_pool*)> p(new gl_prover_pool(), gl_prover_pool_free);
    p->device = device; p->hc = hc;
    const size_t n = hc->hc.n;
    for (uint32_t k = 0; k < lanes; k++) {
        gl_ctx* c = nullptr;
        GL_TRY(gl_ctx_create(device, nullptr, &c));
        p->ctxs.push_back(c);
    }
    GL_TRY(gl_circuit_from_host(p->ctxs[0], hc, &p->circuit));
    p->gens.assign(lanes, nullptr); p->d_wires.assign(lanes, nullptr);
    for (uint32_t k = 0; k < lanes; k++) {
        GL_TRY(gl_matmul_witgen_create(p->ctxs[k], hc, &p->gens[k]));
        void* d = nullptr;
        GL_TRY(gl_dev_alloc(p->ctxs[k], 135 * n * sizeof(uint64_t), &d));
        p->d_wires[k] = (uint64_t*)d;
    }
    *out = p.release();
    return GL_OK;
}

extern "C" uint32_t gl_prover_pool_lanes(const gl_prover_pool* p) { return p ? (uint32_t)p->ctxs.size() : 0; }
extern "C" const gl_circuit* gl_prover_pool_circuit(const gl_prover_pool* p) { return p ? p->circuit : nullptr; }

// count proofs of A_i * B_i = C_i: a[i], b[i] row-major m x m operands on the host, filler_seeds[i] as gl_matmul_witness.
// out_proofs[i] receives a gl_proof (gl_proof_free each); the public inputs are inside the proof bytes.  Returns the first
// error of any lane (proofs already produced stay valid, the others are null).
extern "C" int gl_prover_pool_prove_matmul(gl_prover_pool* p, size_t count, const uint64_t* const* a, const uint64_t* const* b,
                                           const uint64_t* filler_seeds, gl_proof** out_proofs) {
    GL_REQUIRE(p && (count == 0 || (a && b && out_proofs)), GL_ERR_ARG, "gl_prover_pool_prove_matmul: null argument");
    GL_REQUIRE(p->hc, GL_ERR_ARG, "gl_prover_pool_prove_matmul: this pool was created for a generic circuit (gl_prover_pool_prove_columns)");
    for (size_t i = 0; i < count; i++) out_proofs[i] = nullptr;
    const size_t lanes = p->ctxs.size(), npis = 3 * p->hc->hc.m * p->hc->hc.m;
    std::atomic<int> first_error{GL_OK};
    std::vector<std::string> messages(lanes);
    auto work = [&](size_t lane) {
        std::vector<uint64_t> pis(npis);
        uint64_t pi_hash[4];
        for (size_t i = lane; i < count && first_error.load() == GL_OK; i += lanes) {
            int st = (a[i] && b[i]) ? GL_OK : GL_ERR_ARG;
            if (st == GL_OK) st = gl_matmul_witgen_run(p->gens[lane], a[i], b[i], filler_seeds ? filler_seeds[i] : (uint64_t)i, p->d_wires[lane], pis.data(), pi_hash);
            if (st == GL_OK) st = gl_prove_device_hashed(p->ctxs[lane], p->circuit, p->d_wires[lane], pis.data(), npis, pi_hash, &out_proofs[i]);
            if (st != GL_OK) { int expected = GL_OK; messages[lane] = gl_last_error(); first_error.compare_exchange_strong(expected, st); }
        }
    };
    std::vector<std::thread> threads;
    for (size_t k = 1; k < lanes && k < count; k++) threads.emplace_back(work, k);
    work(0);
    for (auto& t : threads) t.join();
    const int st = first_error.load();
    if (st != GL_OK) {
        for (auto& m : messages) if (!m.empty()) return gl_fail(st, m.c_str(), __FILE__, __LINE__);
        return gl_fail(st, "gl_prover_pool_prove_matmul: a lane failed", __FILE__, __LINE__);
    }
    return GL_OK;
}
