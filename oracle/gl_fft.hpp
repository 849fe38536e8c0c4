// ORACLE (test infrastructure only).  CPU restatement of the reference's radix-2 FFT family.
// Follows:
//   field/src/fft.rs:14-33   (fft_root_table)
//   field/src/fft.rs:72-95   (ifft_with_options: forward FFT, then n^-1 scale + index reversal)
//   field/src/fft.rs:142-206 (fft_classic: bit-reverse, zero_factor fill, DIT butterflies)
//   field/src/polynomial/mod.rs:58-88,201-220,282-295 (lde, coset_fft, coset_ifft)
#pragma once
#include "gl_field.hpp"

namespace orc {

typedef std::vector<std::vector<u64>> RootTable;

static inline RootTable fft_root_table(size_t n) {          // fft.rs:14-33
    unsigned lg_n = log2_strict(n);
    std::vector<u64> bases(lg_n);
    u64 base = primitive_root_of_unity(lg_n);
    for (unsigned i = 0; i < lg_n; i++) { bases[i] = base; base = sqr(base); }
    RootTable t;
    for (unsigned lg_m = 1; lg_m <= lg_n; lg_m++) {
        size_t half_m = size_t(1) << (lg_m - 1);
        u64 b = bases[lg_n - lg_m];
        size_t cnt = half_m < 2 ? 2 : half_m;
        std::vector<u64> row(cnt);
        u64 cur = 1;
        for (size_t j = 0; j < cnt; j++) { row[j] = cur; cur = mul(cur, b); }
        t.push_back(row);
    }
    return t;
}

// In-place; `r` = zero_factor: the last (1 - 1/2^r) of the input is known to be zero (fft.rs:169-206).
static inline void fft_classic(u64* v, size_t n, unsigned r, const RootTable& rt) {
    unsigned lg_n = log2_strict(n);
    assert(rt.size() == lg_n);
    for (size_t i = 0; i < n; i++) { size_t j = reverse_bits(i, lg_n); if (i < j) std::swap(v[i], v[j]); }
    if (r > 0) {
        size_t mask = ~((size_t(1) << r) - 1);
        for (size_t i = 0; i < n; i++) v[i] = v[i & mask];
    }
    for (unsigned lg_half_m = r; lg_half_m < lg_n; lg_half_m++) {     // fft.rs:142-160 (scalar packing)
        size_t half_m = size_t(1) << lg_half_m, m = half_m * 2;
        const u64* om = rt[lg_half_m].data();
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half_m; j++) {
                u64 t = mul(om[j], v[k + half_m + j]);
                u64 u = v[k + j];
                v[k + j] = add(u, t);
                v[k + half_m + j] = sub(u, t);
            }
    }
}

static inline void fft_inplace(std::vector<u64>& v, unsigned zero_factor = 0, const RootTable* rt = nullptr) {
    if (v.size() <= 1) return;
    RootTable own;
    if (!rt || rt->size() != log2_strict(v.size())) { own = fft_root_table(v.size()); rt = &own; }
    fft_classic(v.data(), v.size(), zero_factor, *rt);
}

static inline void ifft_inplace(std::vector<u64>& v, const RootTable* rt = nullptr) {   // fft.rs:72-95
    size_t n = v.size();
    if (n <= 1) return;
    unsigned lg_n = log2_strict(n);
    u64 n_inv = inverse_2exp(lg_n);
    fft_inplace(v, 0, rt);
    v[0] = mul(v[0], n_inv);
    v[n / 2] = mul(v[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        u64 ci = mul(v[j], n_inv), cj = mul(v[i], n_inv);
        v[i] = ci; v[j] = cj;
    }
}

// coeffs -> evaluations over shift*H  (polynomial/mod.rs:282-295)
static inline void coset_fft_inplace(std::vector<u64>& c, u64 shift, unsigned zero_factor = 0,
                                     const RootTable* rt = nullptr) {
    u64 pw = 1;
    for (size_t i = 0; i < c.size(); i++) { c[i] = mul(pw, c[i]); pw = mul(pw, shift); }
    fft_inplace(c, zero_factor, rt);
}
// evaluations over shift*H -> coeffs (polynomial/mod.rs:58-70)
static inline void coset_ifft_inplace(std::vector<u64>& v, u64 shift, const RootTable* rt = nullptr) {
    ifft_inplace(v, rt);
    u64 si = inv(shift), pw = 1;
    for (size_t i = 0; i < v.size(); i++) { v[i] = mul(v[i], pw); pw = mul(pw, si); }
}
// zero-pad (PolynomialCoeffs::lde, mod.rs:201-220) then coset FFT with zero_factor = rate_bits
static inline std::vector<u64> lde_coset(const std::vector<u64>& coeffs, unsigned rate_bits, u64 shift,
                                         const RootTable* rt = nullptr) {
    std::vector<u64> p(coeffs);
    p.resize(coeffs.size() << rate_bits, 0);
    coset_fft_inplace(p, shift, rate_bits, rt);
    return p;
}

// Extension-field polynomial FFTs: the reference runs the generic FFT over QuadraticExtension with base
// roots; componentwise base FFTs give the identical result (the transform is F-linear).
static inline void ext_coset_fft_inplace(std::vector<Ext2>& c, u64 shift, unsigned zero_factor = 0) {
    size_t n = c.size();
    std::vector<u64> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = c[i].a; b[i] = c[i].b; }
    RootTable rt = fft_root_table(n);
    coset_fft_inplace(a, shift, zero_factor, &rt);
    coset_fft_inplace(b, shift, zero_factor, &rt);
    for (size_t i = 0; i < n; i++) c[i] = Ext2{a[i], b[i]};
}

// naive O(n^2) evaluation over the subgroup, used by tests as in fft.rs:255-286
static inline std::vector<u64> evaluate_naive(const std::vector<u64>& coeffs) {
    size_t n = coeffs.size(); unsigned lg = log2_strict(n);
    u64 g = primitive_root_of_unity(lg), x = 1;
    std::vector<u64> out(n);
    for (size_t i = 0; i < n; i++) {
        u64 sum = 0, pp = 1;
        for (size_t k = 0; k < n; k++) { sum = add(sum, mul(coeffs[k], pp)); pp = mul(pp, x); }
        out[i] = sum; x = mul(x, g);
    }
    return out;
}

}  // namespace orc
