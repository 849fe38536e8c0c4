// ORACLE (test infrastructure only).  CPU restatement of the reference's PolynomialBatch commit path.
// Follows plonky2/src/fri/oracle.rs:43-133 (from_values, from_coeffs, lde_values, get_lde_values),
// plonky2/src/util/mod.rs:22-28 (transpose) and util/src/lib.rs:188-237 (bit-reversed leaf order).
// Non-zk only (blinding = false; plonk/prover.rs:151, circuit_data.rs:80).
#pragma once
#include "gl_fft.hpp"
#include "gl_poseidon.hpp"
#include <thread>
#include <functional>

namespace orc {

// tiny fork-join helper standing in for Rayon's par_iter over independent items (maybe_rayon/src/lib.rs)
static inline void parallel_for(size_t n, unsigned threads, const std::function<void(size_t)>& fn) {
    if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::vector<std::thread> pool;
    unsigned t = threads < n ? threads : (unsigned)n;
    for (unsigned w = 0; w < t; w++)
        pool.emplace_back([=, &fn]() { for (size_t i = w; i < n; i += t) fn(i); });
    for (auto& th : pool) th.join();
}

struct PolynomialBatch {
    std::vector<std::vector<u64>> polynomials;   // coefficient form, each of length n
    MerkleTree tree;                             // leaves: N x ncols, leaf j = LDE point bitrev(j)
    unsigned degree_log = 0, rate_bits = 0;
    size_t ncols() const { return polynomials.size(); }
    // oracle.rs:128-133
    const u64* get_lde_values(size_t index, size_t step) const {
        size_t idx = reverse_bits(index * step, degree_log + rate_bits);
        return tree.leaf(idx);
    }
};

static inline PolynomialBatch batch_from_coeffs(std::vector<std::vector<u64>> coeffs, unsigned rate_bits,
                                                unsigned cap_height, unsigned threads = 1) {
    PolynomialBatch b;
    size_t ncols = coeffs.size(), n = coeffs[0].size(), N = n << rate_bits;
    b.degree_log = log2_strict(n); b.rate_bits = rate_bits;
    unsigned lgN = b.degree_log + rate_bits;
    RootTable rt = fft_root_table(N);
    std::vector<u64> leaves(N * ncols);
    parallel_for(ncols, threads, [&](size_t c) {                       // oracle.rs:100-125
        assert(coeffs[c].size() == n);
        std::vector<u64> v = lde_coset(coeffs[c], rate_bits, GL_GENERATOR, &rt);
        for (size_t i = 0; i < N; i++) leaves[reverse_bits(i, lgN) * ncols + c] = v[i];   // oracle.rs:83-84
    });
    b.polynomials = std::move(coeffs);
    // Merkle (merkle_tree.rs:135): leaf hashes in parallel, upper levels serial (cheap)
    MerkleTree& t = b.tree;
    t.num_leaves = N; t.leaf_len = ncols; t.cap_height = cap_height; t.leaves = std::move(leaves);
    assert(cap_height <= lgN);
    std::vector<Digest> cur(N);
    parallel_for(threads > 1 ? threads * 8 : 1, threads, [&](size_t w) {
        size_t parts = threads > 1 ? threads * 8 : 1;
        for (size_t i = w * N / parts; i < (w + 1) * N / parts; i++) cur[i] = hash_or_noop(t.leaf(i), ncols);
    });
    t.levels.push_back(cur);
    for (unsigned l = 0; l < lgN - cap_height; l++) {
        std::vector<Digest> nxt(cur.size() / 2);
        size_t parts = (threads > 1 && nxt.size() >= 1024) ? threads * 8 : 1;
        parallel_for(parts, threads, [&](size_t w) {
            for (size_t i = w * nxt.size() / parts; i < (w + 1) * nxt.size() / parts; i++)
                nxt[i] = two_to_one(cur[2 * i], cur[2 * i + 1]);
        });
        t.levels.push_back(nxt);
        cur.swap(nxt);
    }
    return b;
}

static inline PolynomialBatch batch_from_values(std::vector<std::vector<u64>> values, unsigned rate_bits,
                                                unsigned cap_height, unsigned threads = 1) {
    size_t n = values[0].size();
    RootTable rt = fft_root_table(n);
    parallel_for(values.size(), threads, [&](size_t c) { ifft_inplace(values[c], &rt); });   // oracle.rs:51-55
    return batch_from_coeffs(std::move(values), rate_bits, cap_height, threads);
}

}  // namespace orc
