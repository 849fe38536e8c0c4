"""CPU tests: the oracle (oracle/*.hpp) against the reference's known answers and an independent Python model.

These pin the CPU restatement before it is trusted as the checker of the HIP path."""
import numpy as np
import pytest

from oracle_lib import P, rand_field, splitmix64


def ints(a):
    return [int(x) for x in np.asarray(a).reshape(-1)]


def test_field_edge_grid_closed_forms(orc, golden):
    # field/src/prime_field_testing.rs:78-180: every op on the edge grid equals the u128 % p closed form
    g = golden["field_grid"]
    a = np.array([x for x in g for _ in g], dtype=np.uint64)
    b = np.array([y for _ in g for y in g], dtype=np.uint64)
    for op, f in ((0, lambda x, y: (x + y) % P), (1, lambda x, y: (x - y) % P), (2, lambda x, y: (x * y) % P)):
        got = ints(orc.field_op(op, a, b))
        assert got == [f(int(x), int(y)) for x, y in zip(a, b)]
    ga = np.array(g, dtype=np.uint64)
    assert ints(orc.field_op(3, ga)) == [(-int(x)) % P for x in g]
    assert ints(orc.field_op(5, ga)) == [int(x) % P for x in g]
    nz = np.array([x for x in g if x % P], dtype=np.uint64)
    assert ints(orc.field_op(4, nz)) == [pow(int(x), P - 2, P) for x in nz]
    c = np.roll(a, 7)
    assert ints(orc.field_op(6, a, b, c)) == [(int(x) + int(y) * int(z)) % P for x, y, z in zip(a, b, c)]


def test_extension_field(orc):
    a = rand_field(1, 64).reshape(-1, 2)
    b = rand_field(2, 64).reshape(-1, 2)
    got = orc.ext_op(2, a, b).reshape(-1, 2)
    for (a0, a1), (b0, b1), (c0, c1) in zip(a.tolist(), b.tolist(), got.tolist()):
        assert c0 == (a0 * b0 + 7 * a1 * b1) % P and c1 == (a0 * b1 + a1 * b0) % P
    inv = orc.ext_op(3, a)
    one = orc.ext_op(2, a, inv).reshape(-1, 2)
    assert (one[:, 0] == 1).all() and (one[:, 1] == 0).all()


def test_roots_of_unity(orc):
    # field/src/types.rs:268-272, goldilocks_field.rs:87
    for lg in (1, 2, 6, 20, 32):
        w = orc.primitive_root(lg)
        assert pow(w, 1 << lg, P) == 1 and pow(w, 1 << (lg - 1), P) == P - 1
    assert orc.primitive_root(6) == 1 << 39   # 64th root is a power of two: the HIP butterflies rely on it
    assert pow(7, (P - 1) >> 32, P) == 1753635133440165772


def test_fft_reference_test_case(orc, golden):
    # field/src/fft.rs:219-253
    c = np.array(golden["fft256"]["coeffs"], dtype=np.uint64)
    v = orc.fft(c)
    assert ints(v) == golden["fft256"]["values"]
    assert ints(orc.evaluate_naive(c)) == golden["fft256"]["values"]
    assert (orc.ifft(v) == c).all()
    for r in range(4):
        z = np.concatenate([c, np.zeros(256 * ((1 << r) - 1), dtype=np.uint64)])
        assert (orc.fft(z) == orc.fft(z, zero_factor=r)).all()


def test_ntt_python_model(orc, golden):
    for case in golden["ntt_model"]:
        c = np.array(case["coeffs"], dtype=np.uint64)
        assert ints(orc.fft(c)) == case["values"]
        assert ints(orc.coset_fft(c, 7)) == case["coset7"]
        assert (orc.coset_ifft(np.array(case["coset7"], dtype=np.uint64), 7) == c).all()
    m = golden["lde_model"]
    assert ints(orc.lde(np.array(m["coeffs"], dtype=np.uint64), m["rate_bits"])) == m["values"]


def test_bit_reverse_table(golden):
    # plonky2/src/util/mod.rs:57-77
    import oracle_lib  # noqa: F401
    assert golden["bitrev256"] == [int(format(i, "08b")[::-1], 2) for i in range(256)]


def test_poseidon_known_answers(orc, golden):
    # plonky2/src/hash/poseidon_goldilocks.rs:449-485 and poseidon.rs:777-790 (fast == naive)
    for kat in golden["poseidon_kats"] + golden["poseidon_model"]:
        s = np.array(kat["input"], dtype=np.uint64)
        assert ints(orc.poseidon(s)) == kat["output"]
        assert ints(orc.poseidon(s, naive=True)) == kat["output"]
    r = rand_field(3, (64, 12))
    assert (orc.poseidon(r) == orc.poseidon(r, naive=True)).all()


def test_sponge_and_noop(orc):
    # hashing.rs:117-146: overwrite mode, a short last chunk leaves the rest of the rate untouched
    x = splitmix64(5, 11)
    st = np.zeros(12, dtype=np.uint64)
    st[:8] = x[:8]
    st = orc.poseidon(st)
    st[:3] = x[8:]
    st = orc.poseidon(st)
    assert (orc.hash_no_pad(x) == st[:4]).all()
    assert (orc.hash_or_noop(x) == st[:4]).all()
    # config.rs:55-62: <= 4 elements are copied canonically, not hashed
    short = np.array([P + 5, 3, (1 << 64) - 1], dtype=np.uint64)
    assert ints(orc.hash_or_noop(short)) == [5, 3, ((1 << 64) - 1) % P, 0]
    assert ints(orc.hash_no_pad(np.zeros(0, dtype=np.uint64))) == [0, 0, 0, 0]


def test_merkle_python_model(orc, golden):
    m = golden["merkle_model"]
    leaves = np.array(m["leaves"], dtype=np.uint64)
    for t in m["trees"]:
        tree = orc.merkle(leaves, t["cap_height"])
        assert tree.cap.tolist() == t["cap"]
        for idx, sib in t["paths"].items():
            assert tree.prove(int(idx)).tolist() == sib
    s = golden["merkle_short"]
    assert orc.merkle(np.array(s["leaves"], dtype=np.uint64), s["cap_height"]).cap.tolist() == s["cap"]


def test_merkle_every_leaf_verifies(orc):
    # plonky2/src/hash/merkle_tree.rs:223-281: n = 256 leaves x 7, cap heights 1 and 8 (+0, 4)
    leaves = rand_field(9, (256, 7))
    for cap_height in (0, 1, 4, 8):
        tree = orc.merkle(leaves, cap_height)
        cap = tree.cap
        for i in range(256):
            assert orc.merkle_verify(leaves[i], i, cap, tree.prove(i))
        bad = leaves[3].copy()
        bad[0] ^= 1
        assert not orc.merkle_verify(bad, 3, cap, tree.prove(3))


def test_polynomial_batch_python_model(orc, golden):
    m = golden["batch_model"]
    b = orc.batch(np.array(m["values"], dtype=np.uint64), m["rate_bits"], m["cap_height"], from_values=True)
    assert b.polynomials.tolist() == m["coeffs"]
    assert b.cap.tolist() == m["cap"]
    assert b.get_leaf(13).tolist() == m["leaf_13"]
    lde = np.array(m["lde"], dtype=np.uint64)           # [ncols][N] natural order
    leaves = b.leaves()
    for j in range(64):
        assert leaves[j].tolist() == lde[:, int(format(j, "06b")[::-1], 2)].tolist()   # oracle.rs:83-84
    m9 = golden["batch_model9"]
    b9 = orc.batch(np.array(m9["values"], dtype=np.uint64), m9["rate_bits"], m9["cap_height"], from_values=True, threads=2)
    assert b9.cap.tolist() == m9["cap"]
    assert b9.prove(37).tolist() == m9["path_37"]


# ------------------------------------------------------------------------------------------ whole prover
def test_matmul_circuit_shapes(orc):
    # row counts before padding (README.md worked instance): m=2 -> 6 rows, m=20 -> 932 rows
    c2 = orc.circuit(2)
    assert c2.info["degree_bits"] == 3 and c2.info["num_arith_ops"] == 12 and c2.info["num_poseidon_rows"] == 2
    assert c2.info["pi_row"] == 4 and c2.info["constant_row"] == 5           # 2 arithmetic + 2 Poseidon rows first
    assert c2.gate_order() == [0, 1, 2, 3, 4]                                # Noop, Constant, PublicInput, Arithmetic, Poseidon
    assert c2.info["num_constants"] == 4 and c2.info["num_selectors"] == 2   # selectors.rs greedy grouping, max degree 9
    assert c2.info["num_gate_constraints"] == 123 and c2.info["num_partial_products"] == 9
    c20 = orc.circuit(20, threads=4)
    assert c20.info["pi_row"] == 930 and c20.info["constant_row"] == 931 and c20.info["degree_bits"] == 10
    assert c20.info["num_fri_rounds"] == 2 and c20.info["final_poly_len"] == 4
    cs = c2.constants_sigmas()
    # selector columns: group 0 holds gate indices 0..3, UNUSED elsewhere; constants of arithmetic rows are (1,0)/(1,1)
    gates = c2.row_gates().tolist()
    assert gates == [3, 3, 4, 4, 2, 1, 0, 0]
    U = 0xFFFFFFFF
    assert cs[0].tolist() == [3, 3, U, U, 2, 1, 0, 0] and cs[1].tolist() == [U, U, 4, 4, U, U, U, U]
    assert cs[2].tolist()[:2] == [1, 1] and cs[3].tolist()[:2] == [0, 1] and cs[2][5] == 0 and cs[3][5] == 1
    # every sigma column is a permutation image: the multiset {sigma values} equals {k_j * w^i}
    n, w = 8, orc.primitive_root(3)
    ids = sorted(pow(7, j, P) * pow(w, i, P) % P for j in range(80) for i in range(n))
    assert sorted(int(x) for x in cs[4:].reshape(-1)) == ids


def test_readme_instance_m2_proves_and_verifies(orc):
    # README.md:74-84 / matrix_mul.rs:74: A = [[1,2],[3,4]], B = [[5,6],[7,8]]
    c = orc.circuit(2)
    w = c.witness([1, 2, 3, 4], [5, 6, 7, 8])
    pis = w.public_inputs().reshape(-1, 3)
    assert pis.tolist() == [[1, 5, 19], [2, 6, 22], [3, 7, 43], [4, 8, 50]]     # (a_ij, b_ij, c_ij)
    wires = w.wires()
    assert wires.shape == (135, 8)
    assert wires[:, 6].tolist() == [0] * 135 and wires[:, 7].tolist() == [0] * 135   # Noop rows
    proof = w.prove()
    ok, msg = proof.verify()
    assert ok, msg
    assert len(proof.to_bytes()) == 70288
    ch = proof.challenges()
    assert ch["fri_betas"] == [] and len(proof.query_indices()) == 28
    # deterministic: same witness matrix -> same bytes; different filler wires -> different proof, same public inputs
    assert proof.to_bytes() == w.prove().to_bytes()
    w2 = c.witness([1, 2, 3, 4], [5, 6, 7, 8], filler_seed=7)
    p2 = w2.prove()
    assert p2.to_bytes() != proof.to_bytes() and p2.verify()[0]


def test_prover_with_fri_rounds_and_tampering(orc):
    c = orc.circuit(8, threads=4)                      # n = 2^7, one arity-16 FRI round, final poly of 8
    assert c.info["num_fri_rounds"] == 1
    a, b = rand_field(8, 64) % (2**32 - 1), rand_field(9, 64) % (2**32 - 1)
    w = c.witness(a, b)
    A, B = [[int(x) for x in r] for r in a.reshape(8, 8)], [[int(x) for x in r] for r in b.reshape(8, 8)]
    C = [[sum(A[i][k] * B[k][j] for k in range(8)) % P for j in range(8)] for i in range(8)]
    assert w.public_inputs().reshape(64, 3)[:, 2].tolist() == [C[i][j] for i in range(8) for j in range(8)]
    proof = w.prove(threads=4)
    assert proof.verify()[0]
    assert proof.to_bytes() == w.prove(threads=1).to_bytes()       # thread count does not change the proof
    for what, needle in ((0, "vanishing"), (1, "proof of work"), (3, "vanishing"), (4, "Merkle"), (5, "vanishing")):
        bad = w.prove(threads=4)
        bad.tamper(what)
        ok, msg = bad.verify()
        assert not ok and needle in msg, (what, msg)


def test_proof_regression_hashes(orc):
    # Proofs are deterministic functions of (circuit, witness matrix): these SHA-256 values were recorded from the oracle
    # when it was first validated (verifier accepts, GPU byte-identical) and pin BOTH implementations against silent drift,
    # e.g. a change of the PoW choice or of the serialisation order that would still verify.
    import hashlib, json, os
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proof_hashes.json")))
    c = orc.circuit(2)
    assert hashlib.sha256(c.witness([1, 2, 3, 4], [5, 6, 7, 8]).prove().to_bytes()).hexdigest() == want["readme_m2"]
    for m, seed in ((1, 11), (3, 12), (8, 13), (20, 14)):
        oc = orc.circuit(m, threads=8)
        a, b = rand_field(seed, m * m) % (2**32 - 1), rand_field(seed + 100, m * m) % (2**32 - 1)
        pr = oc.witness(a, b, filler_seed=seed).prove(threads=8)
        w = want["m%d_seed%d" % (m, seed)]
        assert [int(x) for x in oc.digest] == w["digest"]
        assert len(pr.to_bytes()) == w["bytes"] and pr.challenges()["pow_witness"] == w["pow_witness"]
        assert hashlib.sha256(pr.to_bytes()).hexdigest() == w["sha256"]


def test_partial_products_reference_vector(orc):
    # the reference's own test (plonky2/src/util/partial_products.rs:114-146): v = [1..6], denominators = 1, Z(x) = 1, Z(gx) = 720
    import ctypes
    lib = orc.lib
    lib.orc_quotient_chunk_products.restype = ctypes.c_size_t
    lib.orc_check_partial_products.restype = ctypes.c_size_t
    u = lambda xs: np.array(xs, dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    v, ones = u([1, 2, 3, 4, 5, 6]), u([1] * 6)
    for max_degree, chunks, pps_z in ((2, [2, 12, 30], [2, 24, 720]), (3, [6, 120], [6, 720])):
        out = np.zeros(8, dtype=np.uint64)
        k = lib.orc_quotient_chunk_products(vp(v), ctypes.c_size_t(6), ctypes.c_size_t(max_degree), vp(out))
        assert out[:k].tolist() == chunks
        acc = np.zeros(k, dtype=np.uint64)
        lib.orc_partial_products_and_z_gx(ctypes.c_uint64(1), vp(u(chunks)), ctypes.c_size_t(k), vp(acc))
        assert acc.tolist() == pps_z
        pps = u(pps_z[:-1])                                   # num_partial_products = ceil(6 / max_degree) - 1
        assert len(pps) == -(-6 // max_degree) - 1
        chk = np.ones(8, dtype=np.uint64)
        c = lib.orc_check_partial_products(vp(v), vp(ones), ctypes.c_size_t(6), vp(pps), ctypes.c_size_t(len(pps)), ctypes.c_uint64(1),
                                           ctypes.c_uint64(720), ctypes.c_size_t(max_degree), vp(chk))
        assert c == len(chunks) and chk[:c].tolist() == [0] * c
        # and a wrong Z(gx) is caught
        lib.orc_check_partial_products(vp(v), vp(ones), ctypes.c_size_t(6), vp(pps), ctypes.c_size_t(len(pps)), ctypes.c_uint64(1),
                                       ctypes.c_uint64(721), ctypes.c_size_t(max_degree), vp(chk))
        assert any(int(x) for x in chk[:c])


# ---------------------------------------------------------------------- Poseidon tables pinned to reference-held data
def _parse_tables(path):
    import re
    text = open(path).read()
    out = {}
    for m in re.finditer(r"(?:POSEIDON_TABLE(?:32)?\((\w+), \d+\)|static const uint64_t (\w+)\[\d+\]) = \{(.*?)\};", text, re.S):
        out[m.group(1) or m.group(2)] = [int(x.rstrip("uUL"), 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+u?\b", m.group(3))]
    return out


def test_poseidon_tables_equal_the_reference_tables():
    """The five FAST_PARTIAL_* tables the reference holds (plonky2/src/hash/poseidon_goldilocks.rs:27-215, committed as data in
    tests/golden/poseidon_fast_tables.json) pin (a) what tools/gen_poseidon_constants.py derives from first principles,
    (b) the product's generated csrc/poseidon_constants.inc and (c) the oracle's own transcribed copy -- the KATs pin only
    the permutation, these tables are what the PoseidonGate constraint code (gates/poseidon.rs:193-272) is written against."""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "poseidon_fast_tables.json")))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_poseidon_constants as g
    rc = g.load_round_constants()
    T = g.derive(rc)
    flat = lambda rows: [x for r in rows for x in r]
    assert T["first"] == ref["FAST_PARTIAL_FIRST_ROUND_CONSTANT"]
    assert T["ks"] == ref["FAST_PARTIAL_ROUND_CONSTANTS"]
    assert flat(T["cols_w"]) == ref["FAST_PARTIAL_ROUND_VS"]
    assert flat(T["rows_v"]) == ref["FAST_PARTIAL_ROUND_W_HATS"]
    assert flat([[T["init"][c][r] for c in range(11)] for r in range(11)]) == ref["FAST_PARTIAL_ROUND_INITIAL_MATRIX"]
    assert g.MDS_CIRC == ref["MDS_MATRIX_CIRC"] and g.MDS_DIAG == ref["MDS_MATRIX_DIAG"]
    want = {"POSEIDON_RC": flat(rc), "POSEIDON_MDS_CIRC": ref["MDS_MATRIX_CIRC"], "POSEIDON_MDS_DIAG": ref["MDS_MATRIX_DIAG"],
            "POSEIDON_PARTIAL_FIRST_RC": ref["FAST_PARTIAL_FIRST_ROUND_CONSTANT"], "POSEIDON_PARTIAL_RC": ref["FAST_PARTIAL_ROUND_CONSTANTS"],
            "POSEIDON_PARTIAL_INIT": ref["FAST_PARTIAL_ROUND_INITIAL_MATRIX"], "POSEIDON_PARTIAL_ROW": ref["FAST_PARTIAL_ROUND_W_HATS"],
            "POSEIDON_PARTIAL_COL": ref["FAST_PARTIAL_ROUND_VS"]}
    product = _parse_tables(os.path.join(root, "plonky2_demo_amd", "csrc", "poseidon_constants.inc"))
    oracle = _parse_tables(os.path.join(root, "oracle", "gl_poseidon_tables.hpp"))
    for name, vals in want.items():
        assert product[name] == vals, "product table %s differs from the reference's" % name
        assert oracle[name] == vals, "oracle table %s differs from the reference's" % name
    # the GPU-only grouped form (three partial rounds at a time) has no reference counterpart: it is validated against the
    # textbook permutation by the generator itself and by the KATs on the device
    G = g.derive_groups(rc, T["M"])
    assert product["POSEIDON_G3_M3"] == flat(G["M3"]) and product["POSEIDON_G3_K"] == flat(G["K"])


@pytest.mark.parametrize("kind,param,inputs,outputs", [
    (3, 99, [0, 1], [0, 1, 3736710860384812976]),            # plonky2/examples/fibonacci.rs: the value its README prints
    (4, 100, [1], [1, 3822706312645553057]),                 # plonky2/examples/factorial.rs
    (5, 0, [1], [1, 4]),                                     # plonky2/examples/easy_polynomial.rs
    (6, 0, [3], [9]),                                        # plonky2/examples/square_root.rs
])
def test_reference_example_circuits_on_the_oracle(orc, kind, param, inputs, outputs):
    # the generic CircuitBuilder restatement (several ConstantGates, the base_arithmetic_results memo) on the reference's own
    # example programs over the five supported gates: public outputs as the examples print them, proof accepted by the verifier
    oc = orc.circuit_of_kind(kind, param, threads=4)
    w = oc.witness(np.array(inputs, dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=3)
    assert [int(x) for x in w.public_inputs()] == outputs
    pr = w.prove(threads=4)
    assert pr.verify()[0]
    if kind == 3:
        assert oc.info["num_arith_ops"] == 99 and oc.info["degree_bits"] == 3      # 5 ArithmeticGates + Poseidon + PI + Constant = 8 rows
    if kind == 4:
        assert oc.info["degree_bits"] == 6      # 99 constants + 0 -> 50 ConstantGates, 5 ArithmeticGates, Poseidon, PI -> 57 rows -> 64


# ---- the reference's own property tests for the pieces of prove_openings, replayed against the oracle -------------------------
def _ext_mul(x, y):                     # (a0 + a1 X)(b0 + b1 X) mod X^2 - 7 with Python integers
    return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def _ext_add(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def test_division_by_linear_property(orc):
    # field/src/polynomial/division.rs:145-158 (there over the quartic extension; the path uses the quadratic one):
    # poly == quotient * (X - z) + poly(z) for random lengths, incl. length 1 (empty quotient)
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 17, 64, int(rng.integers(1, 1000))):
        coeffs = rand_field(1000 + n, (n, 2))
        z = tuple(int(v) for v in rand_field(2000 + n, 2))
        q = [tuple(int(v) for v in row) for row in orc.divide_by_linear(coeffs, z)]
        assert len(q) == n - 1
        ev = (0, 0)
        for c in reversed(coeffs):
            ev = _ext_add(_ext_mul(ev, z), (int(c[0]), int(c[1])))
        negz = ((P - z[0]) % P, (P - z[1]) % P)
        back = [(0, 0)] * n
        for i, qi in enumerate(q):                      # quotient * (X - z)
            back[i] = _ext_add(back[i], _ext_mul(qi, negz))
            back[i + 1] = _ext_add(back[i + 1], qi)
        back[0] = _ext_add(back[0], ev)
        assert back == [(int(c[0]), int(c[1])) for c in coeffs], n


@pytest.mark.parametrize("n", [10, 11, 100])
def test_reducing_factor_properties(orc, n):
    # plonky2/src/util/reducing.rs:248-327 checks ReducingFactor::reduce against its in-circuit twin for n = 10, 11, 100 (the gadget's
    # gates are off this path); the out-of-circuit half is replayed here against the definition: reduce(vs) = sum_j alpha^j v_j,
    # reduce_polys_base = the same coefficient by coefficient with FORWARD powers (:83-94), and the shift by alpha^count (:96-106)
    alpha = tuple(int(v) for v in rand_field(31 + n, 2))
    vs = [(i, 0) for i in range(n)]                                    # test_reduce_gadget: FF::from_canonical_usize(0..n)
    want, ap = (0, 0), (1, 0)
    for v in vs:
        want = _ext_add(want, _ext_mul(ap, v))
        ap = _ext_mul(ap, alpha)
    assert tuple(int(x) for x in orc.reduce_ext(alpha, np.array(vs, dtype=np.uint64))) == want
    base = rand_field(77 + n, n)                                       # test_reduce_gadget_base: random base-field values
    wantb, ap = (0, 0), (1, 0)
    for v in base:
        wantb = _ext_add(wantb, _ext_mul(ap, (int(v), 0)))
        ap = _ext_mul(ap, alpha)
    assert tuple(int(x) for x in orc.reduce_ext(alpha, np.stack([base, np.zeros(n, dtype=np.uint64)], axis=1))) == wantb
    polys = rand_field(99 + n, (n, 8))
    red = orc.reduce_polys_base(alpha, polys)
    for i in range(8):
        assert tuple(int(x) for x in red[i]) == tuple(int(x) for x in orc.reduce_ext(alpha, np.stack([polys[:, i], np.zeros(n, dtype=np.uint64)], axis=1)))
    # after reducing n polynomials the factor's count is n: shifting by alpha^n then adding a second batch is the reduction of the
    # concatenation read backwards -- the "final = alpha^2 Q0 + Q1" order of prove_openings (fri/oracle.rs:193-196)
    two = rand_field(5 + n, (2, 8))
    shift = (1, 0)
    for _ in range(2):
        shift = _ext_mul(shift, alpha)
    red2 = orc.reduce_polys_base(alpha, two)
    both = orc.reduce_polys_base(alpha, np.concatenate([two, polys]))
    for i in range(8):
        lhs = _ext_add(_ext_mul(tuple(int(x) for x in red[i]), shift), tuple(int(x) for x in red2[i]))
        assert lhs == tuple(int(x) for x in both[i])


def test_unique_coset_shifts_give_distinct_cosets(orc):
    # field/src/cosets.rs:33-53: 50 shifts of the subgroup of order 2^5 -- the union of the cosets has no repeated element; and the
    # circuit's k_is (circuit_builder.rs:1007) are the first 80 of them
    shifts = [int(x) for x in orc.unique_coset_shifts(50)]
    assert shifts[:3] == [1, 7, 49]
    g = pow(7, (P - 1) >> 5, P)
    seen = set()
    for sh in shifts:
        x = sh
        for _ in range(32):
            assert x not in seen, "duplicate element"
            seen.add(x)
            x = x * g % P
    assert len(seen) == 50 * 32
    oc = orc.circuit(2, threads=1)
    k = np.zeros(80, dtype=np.uint64)
    orc.lib.orc_circuit_k_is(oc.h, k.ctypes.data_as(__import__("ctypes").c_void_p))
    assert [int(x) for x in k] == [int(x) for x in orc.unique_coset_shifts(80)]
