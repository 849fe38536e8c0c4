"""The product's native verifier (gl_verify, host code as in the reference: plonk/verifier.rs:15-115, fri/verifier.rs:62-260)
against the oracle's restatement: same verdict and same failing check on valid, tampered, mutated and malformed proofs.
Runs without a GPU (proofs come from the CPU oracle prover)."""
import numpy as np
import pytest

from oracle_lib import P, rand_field


def _case(orc, m, seed=0, threads=8):
    import plonky2_demo_amd as p
    oc = orc.circuit(m, threads=threads)
    a, b = rand_field(40 + m + seed, m * m) % (2**32 - 1), rand_field(41 + m + seed, m * m) % (2**32 - 1)
    w = oc.witness(a, b, filler_seed=seed)
    return p.MatmulCircuit(m), oc, w


@pytest.mark.parametrize("m", [1, 2, 3, 5, 8, 20])
def test_accepts_oracle_proofs(orc, m):
    hc, oc, w = _case(orc, m)
    by = w.prove(threads=8).to_bytes()
    ok, why = hc.verify(by, oc.constants_sigmas_cap, oc.digest)
    assert ok, why
    # VerifierOnlyCircuitData is binding: another digest or cap is rejected
    assert not hc.verify(by, oc.constants_sigmas_cap, oc.digest ^ np.uint64(1))[0]
    cap = oc.constants_sigmas_cap.copy()
    cap[:, 2] ^= np.uint64(1)                      # every subtree root: whichever subtrees the 28 queries hit, the path check fails
    ok2, why2 = hc.verify(by, cap, oc.digest)
    assert not ok2 and "initial Merkle proof fails" in why2
    # a proof for other operands does not verify against ... it does (same circuit): soundness is about the statement
    _, _, w2 = _case(orc, m, seed=9)
    assert hc.verify(w2.prove(threads=8).to_bytes(), oc.constants_sigmas_cap, oc.digest)[0]


def test_same_failing_check_as_the_oracle_on_tampered_proofs(orc):
    hc, oc, w = _case(orc, 8)
    for what, needle in ((0, "vanishing"), (1, "proof of work"), (2, "proof of work"), (3, "vanishing"), (4, "Merkle"), (5, "vanishing")):
        bad = w.prove(threads=8)
        bad.tamper(what)
        by = bad.to_bytes()
        ok_o, why_o = oc.verify_bytes(by, oc.constants_sigmas_cap, oc.digest)
        ok_p, why_p = hc.verify(by, oc.constants_sigmas_cap, oc.digest)
        assert not ok_o and not ok_p
        assert needle in why_p and why_p.startswith(why_o), (what, why_o, why_p)


def test_random_mutations_get_the_oracles_verdict(orc):
    hc, oc, w = _case(orc, 8)
    by = w.prove(threads=8).to_bytes()
    rng = np.random.default_rng(5)
    reasons = set()
    for _ in range(60):
        bad = bytearray(by)
        pos = int(rng.integers(0, len(by)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        ok_o, why_o = oc.verify_bytes(bytes(bad), oc.constants_sigmas_cap, oc.digest)
        ok_p, why_p = hc.verify(bytes(bad), oc.constants_sigmas_cap, oc.digest)
        assert ok_o == ok_p, (pos, why_o, why_p)
        if not ok_o and not why_o.startswith("malformed"):
            assert why_p.startswith(why_o), (pos, why_o, why_p)
        reasons.add(why_o.split(" (")[0])
    assert len(reasons) >= 3           # the mutations exercised several different checks
    # a word replaced by the same value + p is the same field element (read_field takes words mod p)
    words = np.frombuffer(by, dtype="<u8").copy() if len(by) % 8 == 0 else None
    if words is not None:
        k = next(i for i in range(200, 400) if int(words[i]) < 2**32 - 1)
        words[k] = np.uint64(int(words[k]) + P)
        assert hc.verify(words.tobytes(), oc.constants_sigmas_cap, oc.digest)[0] == oc.verify_bytes(words.tobytes(), oc.constants_sigmas_cap, oc.digest)[0]


def test_malformed_lengths(orc):
    hc, oc, w = _case(orc, 2)
    by = w.prove().to_bytes()
    for bad in (by[:-1], by[:-8], by + b"\x00", by[: len(by) // 2], b""):
        if not bad:
            continue
        ok, why = hc.verify(bad, oc.constants_sigmas_cap, oc.digest)
        assert not ok and "malformed" in why, why
    with pytest.raises(ValueError):
        hc.verify(by, oc.constants_sigmas_cap[:3], oc.digest)
    # a proof of another circuit size does not parse
    hc3, oc3, w3 = _case(orc, 3)
    ok, why = hc.verify(w3.prove().to_bytes(), oc.constants_sigmas_cap, oc.digest)
    assert not ok


def test_verifier_under_address_and_ub_sanitizers(orc, tmp_path):
    # gl_verify parses untrusted bytes: its host code is rebuilt with -fsanitize=address,undefined (CPU build; GPU sanitizers
    # are not available) and fed the valid proof plus 400 mutations (tools/sanitizer/verify_fuzz.cpp)
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hc, oc, w = _case(orc, 8)
    (tmp_path / "desc.bin").write_bytes(bytes(hc.desc))
    (tmp_path / "cap.bin").write_bytes(oc.constants_sigmas_cap.tobytes())
    (tmp_path / "dig.bin").write_bytes(oc.digest.tobytes())
    (tmp_path / "proof.bin").write_bytes(w.prove(threads=8).to_bytes())
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tools", "sanitizer")])
    r = subprocess.run([os.path.join(root, "tools", "sanitizer", "verify_fuzz"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "valid: 0" in r.stdout and "400 rejected, 0 accepted" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_circuit_data_reader_and_descriptions_under_address_and_ub_sanitizers(orc, tmp_path):
    # gl_common_data_from_bytes / gl_verify_bytes parse untrusted CIRCUIT DATA, and gl_verify / gl_common_data_to_bytes index with every
    # count of a caller-filled description: the host code rebuilt with -fsanitize=address,undefined (tools/sanitizer/data_fuzz.cpp) gets
    # the VerifierCircuitData bytes of the all-gates circuit (two lookup tables, 11 gate types) with 3000 mutations (bit flips,
    # truncations, huge and plausible counts, garbage, inserted bytes) and 3000 mutated descriptions.  (The first run of this harness
    # found gl_verify sizing a vector by an unbounded num_query_rounds.)
    import os, subprocess
    from plonky2_demo_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    oc = orc.circuit_of_kind(15, 9, threads=4)
    w = oc.witness(np.array([200, 300], dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=2)
    desc = oc.product_desc()
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    (tmp_path / "vd.bin").write_bytes(api.verifier_data_to_bytes(desc, cap, dig))
    (tmp_path / "proof.bin").write_bytes(w.prove(threads=4).to_bytes())
    (tmp_path / "desc.bin").write_bytes(bytes(desc))
    (tmp_path / "cap.bin").write_bytes(cap.tobytes())
    (tmp_path / "dig.bin").write_bytes(dig.tobytes())
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tools", "sanitizer"), "data_fuzz"])
    r = subprocess.run([os.path.join(root, "tools", "sanitizer", "data_fuzz"), str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "valid: 0" in r.stdout and "description mutations:" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.parametrize("kind,param", [(1, 16), (1, 5), (1, 40), (2, 50), (2, 400)])
def test_other_circuit_shapes_over_the_same_gate_set(orc, kind, param):
    # circuits the matmul family never produces (oracle/gl_circuit.hpp build_test_circuit): no ArithmeticGate, or no Poseidon
    # gate and NO public inputs with a single selector group, or no NoopGate at all -- gl_verify gets only the descriptor
    import ctypes
    from plonky2_demo_amd._lib import lib, GL_OK, GL_ERR_VERIFY
    oc = orc.circuit_of_kind(kind, param, threads=4)
    a = rand_field(100 * kind + param, param if kind == 1 else 2)
    w = oc.witness(a, np.zeros(0, dtype=np.uint64), filler_seed=kind)
    by = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    buf = np.frombuffer(by, dtype=np.uint8)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    assert lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK, lib.gl_last_error()
    bad = np.frombuffer(by, dtype=np.uint8).copy()
    bad[len(by) // 3] ^= 2
    assert lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(bad), bad.size) == GL_ERR_VERIFY
    assert (kind == 2) == (len(w.public_inputs()) == 0)


# ------------------------------------------------------------------------------------------ circuit data as bytes
# PARITY UNPINNED against the Rust writer: the reference ships no serialised circuit.  What pins these bytes here: the
# product's writer and the oracle's own writer (two restatements of util/serialization/mod.rs:1596-1790,1889-1919) agree byte for
# byte, the reader inverts the writer, and verification through the byte form gives the same verdicts.
@pytest.mark.parametrize("m", [1, 2, 3, 8, 20])
def test_common_and_verifier_data_bytes_match_the_oracle_writer_and_round_trip(orc, m):
    import ctypes
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    hc = p.MatmulCircuit(m)
    oc = orc.circuit(m, threads=4)
    common = api.common_data_to_bytes(hc.desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    assert used == len(common)
    assert bytes(d2) == bytes(hc.desc)                       # every field of the gl_circuit_desc survives the round trip
    assert api.common_data_to_bytes(d2) == common
    cap, dig = oc.constants_sigmas_cap, oc.digest
    vd = api.verifier_data_to_bytes(hc.desc, cap, dig)
    assert vd == oc.data_bytes(1)
    cap2, dig2, used2 = api.verifier_only_from_bytes(vd)
    assert (cap2 == cap).all() and (dig2 == dig).all() and used2 == 8 + 32 * 16 + 32
    # layout facts of the reference's writer: usize = u64 LE (mod.rs:1220-1222), the cap height first (mod.rs:1901)
    assert vd[:8] == (4).to_bytes(8, "little") and common[:8] == (135).to_bytes(8, "little")
    # verification through the byte form: same verdicts as the structured form
    a, b = rand_field(m, m * m) % (2**32 - 1), rand_field(m + 7, m * m) % (2**32 - 1)
    proof = oc.witness(a, b, filler_seed=3).prove(threads=4).to_bytes()
    assert api.verify_bytes(vd, proof) == (True, "")
    bad = bytearray(proof); bad[40] ^= 1
    ok, why = api.verify_bytes(vd, bytes(bad))
    assert not ok and why
    with pytest.raises(p.Plonky2Mi355xError):
        api.verify_bytes(vd[:-1], proof)                     # truncated verifier data
    with pytest.raises(p.Plonky2Mi355xError):
        api.verify_bytes(vd + b"\0", proof)                  # trailing bytes


def test_common_data_reader_rejects_what_the_library_cannot_prove(orc):
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    hc = p.MatmulCircuit(2)
    good = bytearray(api.common_data_to_bytes(hc.desc))
    # the first gate tag sits after config (6 usize + 2 bool + fri config 3 usize + u32 + 1 + 2 usize), fri params (the same fri
    # config + arity vector + degree + bool) and the gate count
    fri = 3 * 8 + 4 + 1 + 16
    off = 6 * 8 + 2 + fri + fri + 8 + 8 * hc.desc.num_fri_rounds + 8 + 1 + 8
    assert int.from_bytes(good[off:off + 4], "little") in (0, 3, 9, 11, 12)
    bad = bytearray(good); bad[off:off + 4] = (13).to_bytes(4, "little")         # RandomAccessGate
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3                                                       # GL_ERR_UNSUPPORTED
    bad = bytearray(good); bad[6 * 8 + 1] = 1                                      # zero_knowledge = true
    with pytest.raises(p.Plonky2Mi355xError):
        api.common_data_from_bytes(bytes(bad))
    bad = bytearray(good); bad[-8:] = (1).to_bytes(8, "little")                    # one lookup table, but no lookup gates / polynomials
    with pytest.raises(p.Plonky2Mi355xError):
        api.common_data_from_bytes(bytes(bad))
    for cut in (0, 7, 60, len(good) - 1):
        with pytest.raises(p.Plonky2Mi355xError):
            api.common_data_from_bytes(bytes(good[:cut]))
    # size fields are never narrowed: num_wires = 2^32 + 135 is refused, not read as 135
    bad = bytearray(good); bad[0:8] = (2**32 + 135).to_bytes(8, "little")
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3
    # fields the description does not carry must equal what the writer emits: security_bits, use_base_arithmetic_gate, the strategy
    bad = bytearray(good); bad[3 * 8:4 * 8] = (96).to_bytes(8, "little")           # security_bits
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3
    bad = bytearray(good); bad[6 * 8] = 0                                          # use_base_arithmetic_gate = false
    with pytest.raises(p.Plonky2Mi355xError):
        api.common_data_from_bytes(bytes(bad))
    strat = 6 * 8 + 2 + 3 * 8 + 4
    assert good[strat] == 1
    bad = bytearray(good); bad[strat + 1:strat + 9] = (3).to_bytes(8, "little")    # ConstantArityBits(3, 5)
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3


def test_common_data_writer_validates_its_description():
    # a public C entry point: an out-of-range description must come back as an error, not as an out-of-bounds read of k_is[80] or a
    # division by zero
    import copy
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    hc = p.MatmulCircuit(2)
    for field, value in (("num_routed_wires", 81), ("num_routed_wires", 2**31), ("quotient_degree_factor", 0), ("num_constants", 1), ("num_gates", 17)):
        d = copy.copy(hc.desc)
        setattr(d, field, value)
        with pytest.raises(p.Plonky2Mi355xError):
            api.common_data_to_bytes(d)
    d = copy.copy(hc.desc)
    d.gate_types[0] = 10                                  # 0..9 are the supported gate types
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_to_bytes(d)
    assert e.value.code == 3
    for nluts, lens in ((1, [2000]), (2, [600, 600]), (5, [1] * 4), (1, [0])):       # must not index past lut[2048] / lut_len[4]
        d = copy.copy(hc.desc)
        d.num_luts, d.num_lookup_polys, d.num_lookup_selectors = nluts, 7, 4 + nluts
        for t, n_ in enumerate(lens):
            d.lut_len[t] = n_
        with pytest.raises(p.Plonky2Mi355xError):
            api.common_data_to_bytes(d)
    d = copy.copy(hc.desc)
    d.gate_params[0] = 1                                    # a table index on a gate that has none
    d.num_luts, d.num_lookup_polys, d.num_lookup_selectors, d.lut_len[0] = 1, 7, 5, 4
    with pytest.raises(p.Plonky2Mi355xError):
        api.common_data_to_bytes(d)


@pytest.mark.parametrize("bits,value", [(6, 42), (70, 2**64 - 2**32 - 5)])
def test_base_sum_gate_circuits_verify_natively_and_through_the_byte_form(orc, bits, value):
    # BaseSumGate<2> (gate type 5): gl_verify accepts the oracle's range_check proofs (plonky2/examples/range_check.rs), rejects tampered
    # ones with the oracle's verdict, and the circuit data round-trips through the reference's byte form (gate tag 2, usize num_limbs)
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    oc = orc.circuit_of_kind(7, bits, threads=4)
    w = oc.witness(np.array([value], dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=1)
    proof = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    assert list(desc.gate_types)[:desc.num_gates].count(5) == 1
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    import ctypes
    from plonky2_demo_amd._lib import lib, GL_OK
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(by):
        buf = np.frombuffer(by, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(proof), lib.gl_last_error()
    common = api.common_data_to_bytes(desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    assert used == len(common) and bytes(d2) == bytes(desc)
    vd = api.verifier_data_to_bytes(desc, cap, dig)
    assert vd == oc.data_bytes(1) and api.verify_bytes(vd, proof) == (True, "")
    rng = np.random.default_rng(bits)
    for _ in range(40):
        bad = bytearray(proof)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]
    # a different limb count is not representable
    i = common.index((2).to_bytes(4, "little") + (63).to_bytes(8, "little"))
    bad = bytearray(common); bad[i + 4:i + 12] = (32).to_bytes(8, "little")
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3


@pytest.mark.parametrize("bits,base,exponent", [(10, 3, 1000), (64, 7, 2**63 + 5), (66, 5, 2**64 - 2**32 - 1), (1, 9, 1), (5, 0, 0)])
def test_exponentiation_gate_circuits_verify_natively_and_through_the_byte_form(orc, bits, base, exponent):
    # ExponentiationGate (gate type 8; gates/exponentiation.rs, 66 power bits from new_from_config, 67 constraints of degree 4) behind
    # CircuitBuilder::exp (gadgets/arithmetic.rs:240-272: the exponent split by a BaseSumGate<2>, unused power bits tied to false):
    # the oracle's proof of base^exponent is accepted by gl_verify, mutations get the oracle's verdict, the circuit data round-trips
    # through the reference's byte form (gate tag 5, usize num_power_bits)
    import ctypes
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    from plonky2_demo_amd._lib import lib, GL_OK
    P = 2**64 - 2**32 + 1
    oc = orc.circuit_of_kind(13, bits, threads=4)
    w = oc.witness(np.array([base, exponent], dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=1)
    assert [int(x) for x in w.public_inputs()] == [base, exponent, pow(base, exponent, P)]
    proof = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    gates = list(desc.gate_types)[:desc.num_gates]
    assert gates.count(8) == 1 and gates.index(8) == len(gates) - 2 and gates[-1] == 4      # degree 4: between Arithmetic (3) and Poseidon (7)
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(by):
        buf = np.frombuffer(by, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(proof), lib.gl_last_error()
    common = api.common_data_to_bytes(desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    assert used == len(common) and bytes(d2) == bytes(desc)
    vd = api.verifier_data_to_bytes(desc, cap, dig)
    assert vd == oc.data_bytes(1) and api.verify_bytes(vd, proof) == (True, "")
    rng = np.random.default_rng(bits)
    for _ in range(40):
        bad = bytearray(proof)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]
    i = common.index((5).to_bytes(4, "little") + (66).to_bytes(8, "little"))
    bad = bytearray(common); bad[i + 4:i + 12] = (32).to_bytes(8, "little")
    with pytest.raises(p.Plonky2Mi355xError) as e:
        api.common_data_from_bytes(bytes(bad))
    assert e.value.code == 3


def cap_proof_circuit_inputs(orc, height, index, seed=11):
    """Inputs of the oracle's Merkle-proof-to-cap circuit (kind 17, cap of four digests): leaf (5), index, siblings, cap (16)."""
    from oracle_lib import rand_field
    leaves = rand_field(seed, (1 << height, 5))
    tree = orc.merkle(leaves, 2)
    a = np.concatenate([leaves[index], np.array([index], dtype=np.uint64), tree.prove(index).reshape(-1), tree.cap.reshape(-1)])
    return a, tree.cap.reshape(-1)


@pytest.mark.parametrize("kind,param", [(16, 1), (16, 2), (16, 3), (16, 4), (16, 5), (16, 6), (17, 2), (17, 6)])
def test_random_access_gate_circuits_verify_natively_and_through_the_byte_form(orc, kind, param):
    # RandomAccessGate::new_from_config(bits) (gate type 9, gate_params = bits in 1..6: 20 / 13 / 8 / 4 / 2 / 1 copies per row, two extra
    # constant wires for bits 2, 4, 5, 6 -- which then host the circuit's constants instead of a ConstantGate) behind
    # CircuitBuilder::random_access (gadgets/random_access.rs:14-47), and verify_merkle_proof_to_cap with a cap of four digests
    # (hash/merkle_proofs.rs:93-150: le_sum of the top index bits, four random accesses): gl_verify accepts the oracle's proofs and agrees
    # with it on mutations; the circuit data round-trips through the byte form (gate tag 13: bits, num_copies, num_extra_constants)
    import ctypes
    import plonky2_demo_amd as p
    from plonky2_demo_amd import api
    from plonky2_demo_amd._lib import lib, GL_OK
    from oracle_lib import rand_field
    oc = orc.circuit_of_kind(kind, param, threads=4)
    if kind == 16:
        v = rand_field(param, (1 << param,))
        idx = np.array([0, (1 << param) - 1, 5 % (1 << param)], dtype=np.uint64)
        w = oc.witness(np.concatenate([v, idx]), np.zeros(0, dtype=np.uint64), filler_seed=1)
        assert [int(x) for x in w.public_inputs()] == [int(i) for i in idx] + [int(v[int(i)]) for i in idx]
        bits = param
    else:
        index = (0x2D5 >> 1) % (1 << param)
        a, cap4 = cap_proof_circuit_inputs(orc, param, index)
        w = oc.witness(a, np.zeros(0, dtype=np.uint64), filler_seed=1)
        assert [int(x) for x in w.public_inputs()] == [int(x) for x in cap4] + [index]
        bits = 2
    proof = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    gates = list(desc.gate_types)[:desc.num_gates]
    assert gates.count(9) == 1 and desc.gate_params[gates.index(9)] == bits
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(by):
        buf = np.frombuffer(by, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(proof), lib.gl_last_error()
    common = api.common_data_to_bytes(desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    assert used == len(common) and bytes(d2) == bytes(desc)
    rng = np.random.default_rng(param)
    for _ in range(30):
        bad = bytearray(proof)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]
    bad = copy_desc = type(desc).from_buffer_copy(bytes(desc))
    bad.gate_params[gates.index(9)] = 7
    with pytest.raises(p.Plonky2Mi355xError):
        api.common_data_to_bytes(bad)


def test_a_circuit_with_every_supported_gate_type_verifies_natively_and_through_the_byte_form(orc):
    # 11 gate types in three selector groups (two tables' LookupGate / LookupTableGate, Noop, Constant, PublicInput, BaseSum<2>, Arithmetic,
    # Exponentiation, Poseidon), 11 constant columns: the filters of several groups with the unused-selector factor (gate.rs:277-284),
    # the group arrays of the description, the byte form of SelectorsInfo with three groups
    import ctypes
    from plonky2_demo_amd import api
    from plonky2_demo_amd._lib import lib, GL_OK
    P = 2**64 - 2**32 + 1
    oc = orc.circuit_of_kind(15, 9, threads=4)
    x, y = 200, 300
    w = oc.witness(np.array([x, y], dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=2)
    s = (7 * x + 1) % 256 + (3 * x * x + 5 * x + 7) % 256
    assert [int(v) for v in w.public_inputs()] == [x, y, s, pow(s + 3, y, P)]
    proof = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    assert desc.num_gates == 11 and desc.num_selectors == 3 and sorted(list(desc.gate_types)[:11]) == [0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8]
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(by):
        buf = np.frombuffer(by, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(proof), lib.gl_last_error()
    common = api.common_data_to_bytes(desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    for t in range(2):
        d2.last_lu_row[t] = desc.last_lu_row[t]
    assert used == len(common) and bytes(d2) == bytes(desc)
    rng = np.random.default_rng(15)
    for _ in range(40):
        bad = bytearray(proof)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]


def merkle_proof_circuit_inputs(orc, height, index, seed=7):
    """Inputs of the oracle's in-circuit Merkle proof verification (kind 14): leaf (5), index, siblings (4 x height), root (4)."""
    from oracle_lib import rand_field
    leaves = rand_field(seed, (1 << height, 5))
    tree = orc.merkle(leaves, 0)
    root = tree.cap[0]
    a = np.concatenate([leaves[index], np.array([index], dtype=np.uint64), tree.prove(index).reshape(-1), root])
    return a, root


@pytest.mark.parametrize("height,index", [(0, 0), (1, 1), (5, 13), (5, 31), (8, 170)])
def test_merkle_proof_circuits_verify_natively(orc, height, index):
    # CircuitBuilder::verify_merkle_proof (hash/merkle_proofs.rs:78-150; its test_recursive_merkle_proof, with a root for the cap): the
    # leaf hash and one SWAPPED PoseidonGate per level, swap = the index bits from a BaseSumGate<2> -- the demo circuit only ever runs the
    # gate with swap = 0.  The oracle's proof is accepted by gl_verify and mutations get the oracle's verdict.
    import ctypes
    from plonky2_demo_amd._lib import lib, GL_OK
    oc = orc.circuit_of_kind(14, height, threads=4)
    a, root = merkle_proof_circuit_inputs(orc, height, index)
    w = oc.witness(a, np.zeros(0, dtype=np.uint64), filler_seed=1)
    assert [int(x) for x in w.public_inputs()] == [int(x) for x in root] + [index]
    proof = w.prove(threads=4).to_bytes()
    desc = oc.product_desc()
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(by):
        buf = np.frombuffer(by, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(proof), lib.gl_last_error()
    rng = np.random.default_rng(height)
    for _ in range(30):
        bad = bytearray(proof)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]


def lookup_outputs(kind, param, inputs):
    """Public inputs the oracle's lookup circuits (oracle/gl_circuit.hpp kinds 8-12) must produce, and their number of tables."""
    P = 2**64 - 2**32 + 1
    big = {i: (3 * i * i + 5 * i + 7) % 256 for i in range(256)}
    if kind == 8:
        return inputs + [big[v] for v in inputs], 1
    if kind == 9:
        t = {1000 + 37 * i: 17 * i * i + 3 for i in range(10)}
        return inputs + [t[v] for v in inputs], 1
    first = {10: {i: (7 * i + 1) % 256 for i in range(256)}, 11: {i: i * i + 1 for i in range(2, 10)}, 12: big}[kind]
    outs = [first[v] for v in inputs[:param]] + [big[v] for v in inputs[param:]]
    return inputs + outs + ([(outs[0] + outs[param]) % P] if kind == 10 else []), (1 if kind == 12 else 2)


@pytest.mark.parametrize("kind,param,inputs", [
    (8, 2, [1, 2]),                                   # lookup_test.rs test_one_lookup: two lookups in a 256-entry table
    (8, 50, list(range(3, 53))),                      # test_many_lookups: two LookupGate rows, the second padded with the table's first entry
    (8, 40, [7] * 40),                                # exactly one full LookupGate row (no padding), one entry looked up 40 times
    (9, 3, [1000, 1037, 1333]),                       # a 10-entry table whose inputs are not their indices
    (10, 2, [3, 200, 17, 255]),                       # test_two_luts: two 256-entry tables, two lookups each, outputs added
    (10, 45, [(7 * i) % 256 for i in range(90)]),     # two tables, two LookupGate rows each
    (11, 3, [2, 9, 5, 0, 128, 255]),                  # test_different_inputs: an 8-entry table on 2..9 next to a 256-entry one
    (12, 2, [1, 2, 3, 4]),                            # test_same_luts: the same table added twice is ONE table
])
def test_lookup_argument_circuits_verify_natively_and_through_the_byte_form(orc, kind, param, inputs):
    # the lookup argument (plonk/prover.rs:425-572, plonk/vanishing_poly.rs:337-670; LookupGate / LookupTableGate, gate types 6 / 7, one
    # of each PER TABLE): the oracle proves (its quotient exists: every lookup constraint vanishes on H), its verifier and the product's independently
    # written gl_verify accept, both give the same verdict on 60 single-bit mutations, the outputs are the table's, and the circuit data
    # (lookup gates with their table, num_lookup_polys / selectors, luts) equals the oracle writer's bytes and round-trips.  PARITY UNPINNED.
    import ctypes
    from plonky2_demo_amd import api
    from plonky2_demo_amd._lib import lib, GL_OK
    oc = orc.circuit_of_kind(kind, param, threads=4)
    w = oc.witness(np.array(inputs, dtype=np.uint64), np.zeros(0, dtype=np.uint64), filler_seed=2)
    pis = [int(x) for x in w.public_inputs()]
    expected, nluts = lookup_outputs(kind, param, inputs)
    assert pis == expected
    pr = w.prove(threads=4)
    assert pr.verify()[0]
    by = pr.to_bytes()
    desc = oc.product_desc()
    assert desc.num_luts == nluts and desc.num_lookup_polys == 7 and desc.num_lookup_selectors == 4 + nluts
    gates = list(desc.gate_types)[:desc.num_gates]
    assert gates.count(6) == nluts and gates.count(7) == nluts
    assert sorted(desc.gate_params[g] for g in range(desc.num_gates) if gates[g] == 7) == list(range(nluts))
    cap, dig = np.ascontiguousarray(oc.constants_sigmas_cap), np.ascontiguousarray(oc.digest)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)

    def native(b):
        buf = np.frombuffer(b, dtype=np.uint8)
        return lib.gl_verify(ctypes.byref(desc), vp(cap), vp(dig), vp(buf), buf.size) == GL_OK
    assert native(by), lib.gl_last_error()
    rng = np.random.default_rng(param)
    for _ in range(60):
        bad = bytearray(by)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        assert native(bytes(bad)) == oc.verify_bytes(bytes(bad), cap, dig)[0]
    common = api.common_data_to_bytes(desc)
    assert common == oc.data_bytes(0)
    d2, used = api.common_data_from_bytes(common)
    assert used == len(common)
    for t in range(nluts):
        assert d2.last_lu_row[t] == 0
        d2.last_lu_row[t] = desc.last_lu_row[t]       # LookupWire.last_lu_gate is prover data: not in CommonCircuitData's bytes
    assert bytes(d2) == bytes(desc)
    vd = api.verifier_data_to_bytes(desc, cap, dig)
    assert vd == oc.data_bytes(1) and api.verify_bytes(vd, by) == (True, "")
