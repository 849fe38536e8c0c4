// ORACLE (test infrastructure only): flat C entry points over the CPU restatement so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
// The product library (libplonky2_mi355x.so) never links or loads this file.
#include "gl_batch.hpp"
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

}  // extern "C"
