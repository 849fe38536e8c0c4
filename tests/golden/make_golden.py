#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Two kinds of data:
 (1) known answers the reference's own tests hold for this path (transcribed data, not code):
     Poseidon width-12 vectors (plonky2/src/hash/poseidon_goldilocks.rs:449-485), the 256-entry bit-reversal
     table (plonky2/src/util/mod.rs:57-77, regenerated here from its definition and spot-checked against the
     rows quoted in the reference test), the field edge-value grid (field/src/prime_field_testing.rs:7-17) and
     the FFT test polynomial (field/src/fft.rs:227-229);
 (2) outputs of an independent Python big-integer model (pow(x, e, p), O(n^2) DFT, textbook Poseidon from
     tools/gen_poseidon_constants.py) for small shapes: NTT / coset-LDE values, Merkle caps and paths,
     PolynomialBatch commitments.  The Rust reference cannot run here (no toolchain), so (2) pins the C++
     oracle and the HIP path against a third, independent implementation of the published algorithm.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
import gen_poseidon_constants as G  # noqa: E402

P = G.P
RC = G.load_round_constants()
M = G.mds_matrix()
G2 = 1753635133440165772


def root(lg):
    return pow(G2, 1 << (32 - lg), P)


def perm(state):
    return G.perm_naive(list(state), RC, M)


def hash_no_pad(xs):
    st = [0] * 12
    for off in range(0, len(xs), 8):
        chunk = xs[off:off + 8]
        st[:len(chunk)] = chunk
        st = perm(st)
    return st[:4]


def hash_or_noop(xs):
    if len(xs) <= 4:
        return [x % P for x in xs] + [0] * (4 - len(xs))
    return hash_no_pad(xs)


def two_to_one(l, r):
    return perm(list(l) + list(r) + [0] * 4)[:4]


def dft(coeffs, shift=1):
    n = len(coeffs)
    lg = n.bit_length() - 1
    w = root(lg)
    return [sum(c * pow(shift * pow(w, i, P) % P, k, P) for k, c in enumerate(coeffs)) % P for i in range(n)]


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def merkle(leaves, cap_height):
    level = [hash_or_noop(l) for l in leaves]
    levels = [level]
    while len(level) > (1 << cap_height):
        level = [two_to_one(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
        levels.append(level)
    return levels


def main():
    rnd = random.Random(0x504C4F4E4B5932)
    out = {}
    out["poseidon_kats"] = [{"input": i, "output": o} for i, o in G.KATS]
    rin = [[rnd.randrange(P) for _ in range(12)] for _ in range(8)]
    rin.append([P - 1 - k for k in range(12)])
    out["poseidon_model"] = [{"input": s, "output": perm(s)} for s in rin]
    out["bitrev256"] = [bitrev(i, 8) for i in range(256)]
    assert out["bitrev256"][:16] == [0x00, 0x80, 0x40, 0xc0, 0x20, 0xa0, 0x60, 0xe0, 0x10, 0x90, 0x50, 0xd0, 0x30, 0xb0, 0x70, 0xf0]
    # field edge grid (prime_field_testing.rs:7-17): word_bits = 64, modulus p
    grid = set()
    for base in (0, 1 << 31, 1 << 32, 1 << 63):
        for d in range(-10, 11):
            v = base + d
            if 0 <= v < (1 << 64):
                grid.add(v)
    for d in range(0, 11):
        grid.add(P - d)
        grid.add((1 << 64) - 1 - d)
    out["field_grid"] = sorted(grid)
    # fft.rs:219-253 polynomial and its naive evaluation
    coeffs = [(i * 1337) % 100 for i in range(200)] + [0] * 56
    out["fft256"] = {"coeffs": coeffs, "values": dft(coeffs)}
    # small NTT / LDE models
    ntt = []
    for lg in (1, 2, 3, 5, 6):
        c = [rnd.randrange(P) for _ in range(1 << lg)]
        ntt.append({"coeffs": c, "values": dft(c), "coset7": dft(c, 7)})
    out["ntt_model"] = ntt
    c = [rnd.randrange(P) for _ in range(8)]
    out["lde_model"] = {"coeffs": c, "rate_bits": 3, "values": dft(c + [0] * 56, 7)}
    # Merkle: 16 leaves x 7 elements, cap heights 0,1,2 + short-leaf (noop) case
    leaves = [[rnd.randrange(P) for _ in range(7)] for _ in range(16)]
    mk = []
    for ch in (0, 1, 2, 4):
        lv = merkle(leaves, ch)
        paths = {}
        for idx in (0, 5, 15):
            sib, i = [], idx
            for l in lv[:-1]:
                sib.append(l[i ^ 1]); i >>= 1
            paths[str(idx)] = sib
        mk.append({"cap_height": ch, "cap": lv[-1], "paths": paths})
    out["merkle_model"] = {"leaves": leaves, "trees": mk}
    short = [[rnd.randrange(1 << 64) for _ in range(3)] for _ in range(8)]
    out["merkle_short"] = {"leaves": short, "cap_height": 1, "cap": merkle(short, 1)[-1]}
    # PolynomialBatch::from_values: 3 columns x n=8, rate_bits 3 (N = 64), cap_height 2
    n, rb = 8, 3
    vals = [[rnd.randrange(P) for _ in range(n)] for _ in range(3)]
    w = root(3)
    ninv = pow(n, P - 2, P)
    coefs = [[sum(v[i] * pow(w, -i * k % n, P) for i in range(n)) * ninv % P for k in range(n)] for v in vals]
    ldes = [dft(cf + [0] * (n * 7), 7) for cf in coefs]
    N = n << rb
    rows = [[ldes[c][bitrev(j, 6)] for c in range(3)] for j in range(N)]
    rows5 = None
    lv = merkle(rows, 2)
    out["batch_model"] = {"values": vals, "rate_bits": rb, "cap_height": 2, "coeffs": coefs, "lde": ldes, "cap": lv[-1],
                          "leaf_13": rows[13]}
    # the same with 9 columns so that leaves take the sponge path with a ragged last chunk
    vals9 = [[rnd.randrange(P) for _ in range(n)] for _ in range(9)]
    coefs9 = [[sum(v[i] * pow(w, -i * k % n, P) for i in range(n)) * ninv % P for k in range(n)] for v in vals9]
    ldes9 = [dft(cf + [0] * (n * 7), 7) for cf in coefs9]
    rows9 = [[ldes9[c][bitrev(j, 6)] for c in range(9)] for j in range(N)]
    lv9 = merkle(rows9, 4)
    sib, i = [], 37
    for l in lv9[:-1]:
        sib.append(l[i ^ 1]); i >>= 1
    out["batch_model9"] = {"values": vals9, "rate_bits": rb, "cap_height": 4, "cap": lv9[-1], "path_37": sib}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote golden.json (%d bytes)" % os.path.getsize(os.path.join(HERE, "golden.json")))


if __name__ == "__main__":
    main()
