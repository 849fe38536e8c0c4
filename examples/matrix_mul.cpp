// The reference's demo program (plonky2/src/bin/matrix_mul.rs, examples/matrix_multiplication.rs) on the MI355X
// backend: "I know A * B = C" for random m x m matrices of u32 entries -- build, prove, print, verify.
//
//   matrix_mul [m = 20] [seed] [batch]  (seed 0 / absent: operands from std::random_device, as the reference draws them
//                                        from ChaChaRng::from_entropy(), matrix_mul.rs:72-80; batch > 1: that many
//                                        independent proofs through gl_prover_pool, four in flight, each verified)
//
// Plain C++ over the C ABI of include/plonky2_mi355x.h; the only GPU-specific line is gl_ctx_create.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/plonky2_mi355x.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int _st = (call);                                                            \
        if (_st != GL_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, _st, gl_last_error()); return 1; } \
    } while (0)

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
    const size_t m = argc > 1 ? (size_t)atoi(argv[1]) : 20;                      // matrix_mul.rs:30
    const uint64_t seed_arg = argc > 2 ? (uint64_t)strtoull(argv[2], nullptr, 10) : 0;
    std::mt19937_64 rng(seed_arg ? seed_arg : ((uint64_t)std::random_device{}() << 32) ^ std::random_device{}());
    const size_t batch = argc > 3 ? (size_t)atoi(argv[3]) : 1;
    // the batch mode keeps several proofs in flight on as many HIP streams; HIP multiplexes streams onto GPU_MAX_HW_QUEUES
    // hardware queues (default 4), so give every stream its own before the runtime initialises
    setenv("GPU_MAX_HW_QUEUES", "16", 0);

    // ---- build (matrix_mul.rs:25-67): circuit description on the host, constants/sigmas commitment on the GPU ----
    auto t0 = std::chrono::steady_clock::now();
    gl_host_circuit* hc = nullptr;
    CHECK(gl_matmul_circuit_build(m, &hc));
    gl_circuit_desc desc;
    CHECK(gl_host_circuit_desc(hc, &desc));
    gl_ctx* ctx = nullptr;
    CHECK(gl_ctx_create(0, nullptr, &ctx));
    gl_circuit* circuit = nullptr;
    CHECK(gl_circuit_from_host(ctx, hc, &circuit));
    CHECK(gl_circuit_warm_up(ctx, circuit));          // part of build(): kernels loaded, twiddle tables built, pool sized (the reference
                                                      // precomputes its fft_root_table in build() as well, circuit_builder.rs:1016-1019)
    const size_t n = size_t(1) << desc.degree_bits, ncap = size_t(1) << desc.cap_height;
    std::vector<uint64_t> cap(4 * ncap), digest(4);
    CHECK(gl_circuit_constants_sigmas_cap(circuit, cap.data()));
    CHECK(gl_circuit_digest(circuit, digest.data()));
    fprintf(stderr, "build: m = %zu, %zu rows (2^%u), %.1f ms\n", m, n, desc.degree_bits, ms_since(t0));

    // ---- witness (matrix_mul.rs:70-83): u32 operands, gen_range(u32::MIN..u32::MAX) ----
    std::vector<uint64_t> a(m * m), b(m * m), pis(3 * m * m), pi_hash(4);
    std::uniform_int_distribution<uint64_t> u32(0, 0xFFFFFFFEull);
    for (auto& x : a) x = u32(rng);
    for (auto& x : b) x = u32(rng);
    t0 = std::chrono::steady_clock::now();
    gl_matmul_witgen* gen = nullptr;
    CHECK(gl_matmul_witgen_create(ctx, hc, &gen));
    void* d_wires = nullptr;
    CHECK(gl_dev_alloc(ctx, 135 * n * sizeof(uint64_t), &d_wires));
    CHECK(gl_matmul_witgen_run(gen, a.data(), b.data(), rng(), (uint64_t*)d_wires, pis.data(), pi_hash.data()));
    fprintf(stderr, "witness: %.1f ms\n", ms_since(t0));

    // ---- prove (matrix_mul.rs:86) ----
    t0 = std::chrono::steady_clock::now();
    gl_proof* proof = nullptr;
    CHECK(gl_prove_device_hashed(ctx, circuit, (const uint64_t*)d_wires, pis.data(), pis.size(), pi_hash.data(), &proof));
    std::vector<uint8_t> bytes(gl_proof_num_bytes(proof));
    CHECK(gl_proof_bytes(proof, bytes.data(), bytes.size()));
    fprintf(stderr, "prove: %.1f ms, proof of %zu bytes\n", ms_since(t0), bytes.size());

    printf("length of proof.public_inputs is %zu\n", pis.size());                 // matrix_mul.rs:90

    // ---- verify (matrix_mul.rs:106) ----
    t0 = std::chrono::steady_clock::now();
    int st = gl_verify(&desc, cap.data(), digest.data(), bytes.data(), bytes.size());
    fprintf(stderr, "verify: %.1f ms: %s\n", ms_since(t0), st == GL_OK ? "accepted" : gl_last_error());

    // ---- optional: a batch of independent proofs, several in flight (what a Rayon pool does for the reference) ----
    if (st == GL_OK && batch > 1) {
        gl_prover_pool* pool = nullptr;
        CHECK(gl_prover_pool_create(0, hc, 4, &pool));
        std::vector<std::vector<uint64_t>> as(batch, std::vector<uint64_t>(m * m)), bs(batch, std::vector<uint64_t>(m * m));
        std::vector<const uint64_t*> pa(batch), pb(batch);
        for (size_t i = 0; i < batch; i++) {
            for (auto& x : as[i]) x = u32(rng);
            for (auto& x : bs[i]) x = u32(rng);
            pa[i] = as[i].data(); pb[i] = bs[i].data();
        }
        std::vector<gl_proof*> proofs(batch, nullptr);
        t0 = std::chrono::steady_clock::now();
        CHECK(gl_prover_pool_prove_matmul(pool, batch, pa.data(), pb.data(), nullptr, proofs.data()));
        const double ms = ms_since(t0);
        fprintf(stderr, "batch: %zu proofs in %.1f ms = %.1f proofs/s (witness generation included)\n", batch, ms, batch * 1e3 / ms);
        for (size_t i = 0; i < batch && st == GL_OK; i++) {
            std::vector<uint8_t> by(gl_proof_num_bytes(proofs[i]));
            CHECK(gl_proof_bytes(proofs[i], by.data(), by.size()));
            st = gl_verify(&desc, cap.data(), digest.data(), by.data(), by.size());
        }
        fprintf(stderr, "batch verify: %s\n", st == GL_OK ? "all accepted" : gl_last_error());
        for (auto* pr : proofs) gl_proof_free(pr);
        gl_prover_pool_free(pool);
    }

    gl_proof_free(proof);
    gl_matmul_witgen_free(gen);
    gl_dev_free(ctx, d_wires);
    gl_circuit_free(circuit);
    gl_ctx_destroy(ctx);
    gl_host_circuit_free(hc);
    return st == GL_OK ? 0 : 1;
}
