// ORACLE (test infrastructure only -- never linked into the product library).
// CPU restatement of the reference's Goldilocks field, quadratic extension and helper
// functions.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Follows (reference file:line):
//   field/src/goldilocks_field.rs:13,138-142,152,170-178,186-274,346-403   (add/sub/mul/reduce)
//   field/src/types.rs:133-223 (batch inverse), :227-266 (inverse_2exp), :268-283 (roots),
//                      :367-378 (exp_u64), :437-439 (coset shift)
//   field/src/extension/quadratic.rs:143-193, field/src/goldilocks_extensions.rs:14-39 (W = 7)
//   field/src/inversion.rs (inverse; here by Fermat, same value)
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 GL_P = 0xFFFFFFFF00000001ULL;
static const u64 GL_EPS = 0xFFFFFFFFULL;                 // 2^32 - 1 == 2^64 mod p
static const u64 GL_GENERATOR = 7;                        // goldilocks_field.rs:80
static const u64 GL_POW2_GENERATOR = 1753635133440165772ULL;  // goldilocks_field.rs:87
static const unsigned GL_TWO_ADICITY = 32;

// All functions accept possibly non-canonical u64 (any value < 2^64) and return a value that is
// congruent mod p but possibly non-canonical, exactly as the reference does; `canon` gives the
// canonical representative (goldilocks_field.rs:170-178).
static inline u64 canon(u64 x) { return x >= GL_P ? x - GL_P : x; }

static inline u64 add(u64 a, u64 b) {            // goldilocks_field.rs:199-221
    u64 s = a + b;
    bool over = s < a;
    u64 s2 = s + (over ? GL_EPS : 0);
    bool over2 = s2 < s;
    if (over2) s2 += GL_EPS;
    return s2;
}
static inline u64 sub(u64 a, u64 b) {            // goldilocks_field.rs:236-258
    u64 d = a - b;
    bool under = a < b;
    u64 d2 = d - (under ? GL_EPS : 0);
    bool under2 = under && d < GL_EPS;
    if (under2) d2 -= GL_EPS;
    return d2;
}
static inline u64 neg(u64 a) { u64 c = canon(a); return c == 0 ? 0 : GL_P - c; }  // :186-197

static inline u64 add_no_canon(u64 x, u64 y) {   // goldilocks_field.rs:336-342 (x + y < 2^64 + p)
    u64 r = x + y;
    return r + (r < x ? GL_EPS : 0);
}
static inline u64 reduce96(u64 lo, uint32_t hi) {   // goldilocks_field.rs:346-351
    return add_no_canon(lo, (u64)hi * GL_EPS);
}
static inline u64 reduce128(u128 x) {              // goldilocks_field.rs:355-369
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    return add_no_canon(t0, hl * GL_EPS);
}
static inline u64 mul(u64 a, u64 b) { return reduce128((u128)a * b); }   // :267-274
static inline u64 mul_add(u64 acc, u64 x, u64 y) { return reduce128((u128)acc + (u128)x * y); }  // :138-142
static inline u64 sqr(u64 a) { return mul(a, a); }

static inline u64 exp_u64(u64 base, u64 e) {       // types.rs:367-378
    u64 cur = base, prod = 1;
    while (e) { if (e & 1) prod = mul(prod, cur); cur = sqr(cur); e >>= 1; }
    return prod;
}
static inline u64 exp_pow2(u64 base, unsigned k) { while (k--) base = sqr(base); return base; }
static inline u64 inv(u64 a) { assert(canon(a) != 0); return exp_u64(a, GL_P - 2); }

static inline u64 inverse_2exp(unsigned e) {       // types.rs:227-266 (e <= 32 is all we need)
    assert(e <= GL_TWO_ADICITY);
    return GL_P - ((GL_P - 1) >> e);
}
static inline u64 primitive_root_of_unity(unsigned n_log) {   // types.rs:268-272
    assert(n_log <= GL_TWO_ADICITY);
    return exp_pow2(GL_POW2_GENERATOR, GL_TWO_ADICITY - n_log);
}

// Montgomery batch inversion; any valid algorithm gives identical (unique) inverses (types.rs:133-223).
static inline std::vector<u64> batch_inverse(const std::vector<u64>& x) {
    size_t n = x.size();
    std::vector<u64> pre(n), out(n);
    u64 acc = 1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = mul(acc, x[i]); }
    u64 ai = inv(acc);
    for (size_t i = n; i-- > 0;) { out[i] = mul(ai, pre[i]); ai = mul(ai, x[i]); }
    return out;
}

// ---- quadratic extension F[X]/(X^2 - 7) (quadratic.rs:14,143-193; goldilocks_extensions.rs:14-39)
struct Ext2 {
    u64 a, b;   // a + b*X
};
static const u64 EXT_W = 7;
static inline Ext2 ext(u64 a, u64 b = 0) { return Ext2{a, b}; }
static inline Ext2 eadd(Ext2 x, Ext2 y) { return Ext2{add(x.a, y.a), add(x.b, y.b)}; }
static inline Ext2 esub(Ext2 x, Ext2 y) { return Ext2{sub(x.a, y.a), sub(x.b, y.b)}; }
static inline Ext2 eneg(Ext2 x) { return Ext2{neg(x.a), neg(x.b)}; }
static inline Ext2 emul(Ext2 x, Ext2 y) {
    u64 c0 = add(mul(x.a, y.a), mul(EXT_W, mul(x.b, y.b)));
    u64 c1 = add(mul(x.a, y.b), mul(x.b, y.a));
    return Ext2{c0, c1};
}
static inline Ext2 escalar(Ext2 x, u64 s) { return Ext2{mul(x.a, s), mul(x.b, s)}; }   // extension/mod.rs:102-108
static inline Ext2 ecanon(Ext2 x) { return Ext2{canon(x.a), canon(x.b)}; }
static inline bool eeq(Ext2 x, Ext2 y) { return canon(x.a) == canon(y.a) && canon(x.b) == canon(y.b); }
static inline Ext2 einv(Ext2 x) {               // quadratic.rs:100-116 (Frobenius): 1/(a+bX) = (a-bX)/(a^2-7b^2)
    u64 norm = sub(sqr(x.a), mul(EXT_W, sqr(x.b)));
    u64 ni = inv(norm);
    return Ext2{mul(x.a, ni), mul(neg(x.b), ni)};
}
static inline Ext2 eexp_u64(Ext2 base, u64 e) {
    Ext2 cur = base, prod = ext(1);
    while (e) { if (e & 1) prod = emul(prod, cur); cur = emul(cur, cur); e >>= 1; }
    return prod;
}
static inline Ext2 eexp_pow2(Ext2 b, unsigned k) { while (k--) b = emul(b, b); return b; }

// ---- bit utilities (util/src/lib.rs:30-40; plonky2/src/util/mod.rs:30-38)
static inline unsigned log2_strict(size_t n) {
    unsigned l = 0; while ((size_t(1) << l) < n) l++;
    assert((size_t(1) << l) == n);
    return l;
}
static inline unsigned log2_ceil(size_t n) { unsigned l = 0; while ((size_t(1) << l) < n) l++; return l; }
static inline size_t reverse_bits(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); }
    return r;
}
template <class T>
static inline void reverse_index_bits_in_place(std::vector<T>& v) {   // util/src/lib.rs:188-237 (result only)
    size_t n = v.size(); unsigned lg = log2_strict(n);
    for (size_t i = 0; i < n; i++) { size_t j = reverse_bits(i, lg); if (i < j) std::swap(v[i], v[j]); }
}

}  // namespace orc
