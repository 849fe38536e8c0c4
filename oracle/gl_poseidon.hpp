// ORACLE (test infrastructure only).  CPU restatement of the reference's Poseidon permutation,
// sponge, two-to-one compression and Merkle tree.
// Follows:
//   plonky2/src/hash/poseidon.rs:176-198,240-261 (MDS with lazy u128 accumulation + reduce96)
//   plonky2/src/hash/poseidon.rs:311-366,399-427,482-493,521-550,573-609 (round structure)
//   plonky2/src/hash/hashing.rs:98-146 (compress, hash_n_to_m_no_pad: overwrite-mode sponge)
//   plonky2/src/plonk/config.rs:55-66 (hash_or_noop)
//   plonky2/src/hash/merkle_tree.rs:69-207, merkle_proofs.rs:54-75 (tree, cap, prove, verify)
// Constant tables: gl_poseidon_tables.hpp, the oracle's OWN copy transcribed from reference-held data (oracle/make_tables.py).
#pragma once
#include "gl_field.hpp"
#include "gl_poseidon_tables.hpp"
#include <array>
#include <cstring>

namespace orc {

typedef std::array<u64, 12> PState;
struct Digest {
    u64 e[4];
};
static inline bool digest_eq(const Digest& x, const Digest& y) {
    for (int i = 0; i < 4; i++) if (canon(x.e[i]) != canon(y.e[i])) return false;
    return true;
}

static inline u64 sbox7(u64 x) { u64 x2 = sqr(x), x4 = sqr(x2), x3 = mul(x, x2); return mul(x3, x4); }   // poseidon.rs:521-528

static inline void mds_layer(PState& s) {                      // poseidon.rs:176-198,240-261
    PState out;
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * POSEIDON_MDS_CIRC[i];
        acc += (u128)s[r] * POSEIDON_MDS_DIAG[r];
        out[r] = reduce96((u64)acc, (uint32_t)(acc >> 64));
    }
    s = out;
}
static inline void full_round(PState& s, int round) {          // poseidon.rs:573-581
    for (int i = 0; i < 12; i++) s[i] = sbox7(add(s[i], POSEIDON_RC[12 * round + i]));
    mds_layer(s);
}
static inline void partial_rounds(PState& s) {                 // poseidon.rs:583-596
    for (int i = 0; i < 12; i++) s[i] = add(s[i], POSEIDON_PARTIAL_FIRST_RC[i]);
    {   // dense pre-matrix (poseidon.rs:338-366)
        PState t; t[0] = s[0];
        for (int c = 1; c < 12; c++) {
            u64 acc = 0;
            for (int r = 1; r < 12; r++) acc = add(acc, mul(s[r], POSEIDON_PARTIAL_INIT[(r - 1) * 11 + (c - 1)]));
            t[c] = acc;
        }
        s = t;
    }
    const u64 m00 = POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0];
    for (int r = 0; r < POSEIDON_PARTIAL_ROUNDS; r++) {         // poseidon.rs:399-427
        u64 s0 = add(sbox7(s[0]), POSEIDON_PARTIAL_RC[r]);
        u64 d = mul(s0, m00);
        for (int i = 1; i < 12; i++) d = add(d, mul(s[i], POSEIDON_PARTIAL_ROW[r * 11 + i - 1]));
        for (int i = 1; i < 12; i++) s[i] = mul_add(s[i], s0, POSEIDON_PARTIAL_COL[r * 11 + i - 1]);
        s[0] = d;
    }
}
static inline void poseidon(PState& s) {                       // poseidon.rs:598-609
    int round = 0;
    for (int i = 0; i < 4; i++) full_round(s, round++);
    partial_rounds(s);
    round += POSEIDON_PARTIAL_ROUNDS;
    for (int i = 0; i < 4; i++) full_round(s, round++);
}
// textbook form (poseidon.rs:611-633), used to cross-check the fast partial rounds
static inline void poseidon_naive(PState& s) {
    int round = 0;
    for (int i = 0; i < 4; i++) full_round(s, round++);
    for (int k = 0; k < POSEIDON_PARTIAL_ROUNDS; k++, round++) {
        for (int i = 0; i < 12; i++) s[i] = add(s[i], POSEIDON_RC[12 * round + i]);
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int i = 0; i < 4; i++) full_round(s, round++);
}

// hashing.rs:117-146 with num_outputs = 4
static inline Digest hash_no_pad(const u64* in, size_t n) {
    PState st; st.fill(0);
    for (size_t off = 0; off < n; off += 8) {
        size_t c = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < c; i++) st[i] = in[off + i];
        poseidon(st);
    }
    return Digest{{st[0], st[1], st[2], st[3]}};
}
// config.rs:55-66: short inputs are copied (canonicalised through the byte round trip), not hashed
static inline Digest hash_or_noop(const u64* in, size_t n) {
    if (n <= 4) {
        Digest d{{0, 0, 0, 0}};
        for (size_t i = 0; i < n; i++) d.e[i] = canon(in[i]);
        return d;
    }
    return hash_no_pad(in, n);
}
static inline Digest two_to_one(const Digest& l, const Digest& r) {   // hashing.rs:98-115
    PState st; st.fill(0);
    for (int i = 0; i < 4; i++) { st[i] = l.e[i]; st[4 + i] = r.e[i]; }
    poseidon(st);
    return Digest{{st[0], st[1], st[2], st[3]}};
}

// Merkle tree with a cap.  The reference's interleaved digest array (merkle_tree.rs:43-51) is an
// unobservable layout detail; here levels[0] = leaf digests, levels[k] = level k above them, the last
// level is the cap (2^cap_height digests).  cap, prove() and verify match merkle_tree.rs:135-207.
struct MerkleTree {
    size_t num_leaves = 0, leaf_len = 0;
    unsigned cap_height = 0;
    std::vector<u64> leaves;                  // row-major num_leaves x leaf_len
    std::vector<std::vector<Digest>> levels;  // levels[0].size() == num_leaves ... levels.back() == cap
    const std::vector<Digest>& cap() const { return levels.back(); }
    const u64* leaf(size_t i) const { return leaves.data() + i * leaf_len; }
    std::vector<Digest> prove(size_t leaf_index) const {
        std::vector<Digest> sib;
        size_t idx = leaf_index;
        for (size_t l = 0; l + 1 < levels.size(); l++) { sib.push_back(levels[l][idx ^ 1]); idx >>= 1; }
        return sib;
    }
};
static inline MerkleTree merkle_build(std::vector<u64> leaves, size_t num_leaves, size_t leaf_len, unsigned cap_height) {
    unsigned lg = log2_strict(num_leaves);
    assert(cap_height <= lg);                                  // merkle_tree.rs:137-143
    MerkleTree t; t.num_leaves = num_leaves; t.leaf_len = leaf_len; t.cap_height = cap_height;
    t.leaves = std::move(leaves);
    std::vector<Digest> cur(num_leaves);
    for (size_t i = 0; i < num_leaves; i++) cur[i] = hash_or_noop(t.leaf(i), leaf_len);
    t.levels.push_back(cur);
    for (unsigned l = 0; l < lg - cap_height; l++) {
        std::vector<Digest> nxt(cur.size() / 2);
        for (size_t i = 0; i < nxt.size(); i++) nxt[i] = two_to_one(cur[2 * i], cur[2 * i + 1]);
        t.levels.push_back(nxt);
        cur.swap(nxt);
    }
    return t;
}
static inline bool merkle_verify(const u64* leaf, size_t leaf_len, size_t leaf_index,
                                 const std::vector<Digest>& cap, const std::vector<Digest>& siblings) {
    size_t index = leaf_index;                                 // merkle_proofs.rs:54-75
    Digest cur = hash_or_noop(leaf, leaf_len);
    for (const Digest& s : siblings) {
        cur = (index & 1) ? two_to_one(s, cur) : two_to_one(cur, s);
        index >>= 1;
    }
    return index < cap.size() && digest_eq(cur, cap[index]);
}

}  // namespace orc
