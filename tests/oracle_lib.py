"""TEST-ONLY loader for the CPU oracle (oracle/liboracle.so).  Never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001

_sz = ctypes.c_size_t
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_vp = ctypes.c_void_p


def _p(a):
    return a.ctypes.data_as(_vp)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_primitive_root.restype = _u64
        lib.orc_inverse_2exp.restype = _u64
        lib.orc_merkle_new.restype = _vp
        lib.orc_batch_new.restype = _vp
        lib.orc_merkle_prove.restype = _sz
        lib.orc_batch_prove.restype = _sz
        lib.orc_merkle_num_levels.restype = _sz
        lib.orc_batch_num_levels.restype = _sz

    # ---- field
    def field_op(self, op, a, b=None, c=None):
        a = u64(a)
        b = u64(b) if b is not None else a
        c = u64(c) if c is not None else a
        out = np.empty_like(a)
        self.lib.orc_field_op(op, _p(a), _p(b), _p(c), _p(out), _sz(a.size))
        return out

    def ext_op(self, op, a, b=None):
        a = u64(a)
        out = np.empty_like(a)
        self.lib.orc_ext_op(op, _p(a), _p(u64(b)) if b is not None else None, _p(out), _sz(a.size // 2))
        return out

    def primitive_root(self, lg):
        return int(self.lib.orc_primitive_root(_u32(lg)))

    # ---- transforms on [n] or [batch][n]
    def _tf(self, fn, a, *extra):
        a = u64(a).copy()
        n = a.shape[-1]
        batch = a.size // n
        fn(_p(a), _sz(n), _sz(batch), *extra)
        return a

    def fft(self, a, zero_factor=0):
        return self._tf(self.lib.orc_fft, a, _u32(zero_factor))

    def ifft(self, a):
        return self._tf(self.lib.orc_ifft, a)

    def coset_fft(self, a, shift=7, zero_factor=0):
        return self._tf(self.lib.orc_coset_fft, a, _u64(shift), _u32(zero_factor))

    def coset_ifft(self, a, shift=7):
        return self._tf(self.lib.orc_coset_ifft, a, _u64(shift))

    def lde(self, coeffs, rate_bits, threads=1):
        c = u64(coeffs)
        single = c.ndim == 1
        c2 = c.reshape(1, -1) if single else c
        batch, n = c2.shape
        out = np.empty((batch, n << rate_bits), dtype=np.uint64)
        self.lib.orc_lde(_p(c2), _sz(n), _sz(batch), _u32(rate_bits), _p(out), _u32(threads))
        return out[0] if single else out

    def evaluate_naive(self, coeffs):
        c = u64(coeffs)
        out = np.empty_like(c)
        self.lib.orc_evaluate_naive(_p(c), _sz(c.size), _p(out))
        return out

    # ---- stand-alone pieces of prove_openings (util/reducing.rs, polynomial/division.rs, cosets.rs)
    def reduce_polys_base(self, alpha, polys):
        """sum_j alpha^j polys[j]: alpha = (a0, a1), polys [npolys][n] base-field coefficients -> [n][2]."""
        ps = np.ascontiguousarray(u64(polys))
        out = np.empty((ps.shape[1], 2), dtype=np.uint64)
        self.lib.orc_reduce_polys_base(_p(u64(alpha)), _p(ps), _sz(ps.shape[0]), _sz(ps.shape[1]), _p(out))
        return out

    def reduce_ext(self, alpha, vals):
        v = np.ascontiguousarray(u64(vals)).reshape(-1, 2)
        out = np.empty(2, dtype=np.uint64)
        self.lib.orc_reduce_ext(_p(u64(alpha)), _p(v), _sz(v.shape[0]), _p(out))
        return out

    def divide_by_linear(self, coeffs, z):
        c = np.ascontiguousarray(u64(coeffs)).reshape(-1, 2)
        out = np.empty((max(c.shape[0] - 1, 0), 2), dtype=np.uint64)
        self.lib.orc_divide_by_linear(_p(c), _sz(c.shape[0]), _p(u64(z)), _p(out))
        return out

    def unique_coset_shifts(self, num):
        out = np.empty(num, dtype=np.uint64)
        self.lib.orc_unique_coset_shifts(_sz(num), _p(out))
        return out

    # ---- hashing
    def poseidon(self, states, naive=False):
        s = u64(states).copy()
        self.lib.orc_poseidon(_p(s), _sz(s.size // 12), 1 if naive else 0)
        return s

    def hash_or_noop(self, row):
        r = u64(row)
        out = np.empty(4, dtype=np.uint64)
        self.lib.orc_hash_or_noop(_p(r), _sz(r.size), _p(out))
        return out

    def hash_no_pad(self, row):
        r = u64(row)
        out = np.empty(4, dtype=np.uint64)
        self.lib.orc_hash_no_pad(_p(r), _sz(r.size), _p(out))
        return out

    def two_to_one(self, l, r):
        out = np.empty(4, dtype=np.uint64)
        self.lib.orc_two_to_one(_p(u64(l)), _p(u64(r)), _p(out))
        return out

    def merkle(self, leaves, cap_height):
        return OracleMerkle(self, leaves, cap_height)

    def merkle_verify(self, leaf, index, cap, siblings):
        leaf, cap, siblings = u64(leaf), u64(cap), u64(siblings)
        return bool(self.lib.orc_merkle_verify(_p(leaf), _sz(leaf.size), _sz(index), _p(cap), _sz(cap.size // 4),
                                               _p(siblings), _sz(siblings.size // 4)))

    def batch(self, cols, rate_bits, cap_height, from_values=True, threads=1):
        return OracleBatch(self, cols, rate_bits, cap_height, from_values, threads)

    def circuit(self, m, threads=1, verifier_only=False):
        return OracleCircuit(self, m, threads, verifier_only)

    def circuit_of_kind(self, kind, param, threads=1):
        return OracleCircuit.of_kind(self, kind, param, threads)


class OracleMerkle:
    def __init__(self, o, leaves, cap_height):
        self.o = o
        l2 = u64(leaves)
        self.num_leaves, self.leaf_len = l2.shape
        self.cap_height = cap_height
        self.h = _vp(o.lib.orc_merkle_new(_p(l2), _sz(self.num_leaves), _sz(self.leaf_len), _u32(cap_height)))

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.o.lib.orc_merkle_cap(self.h, _p(out))
        return out

    def prove(self, i):
        out = np.empty((64, 4), dtype=np.uint64)
        n = self.o.lib.orc_merkle_prove(self.h, _sz(i), _p(out))
        return out[:n].copy()

    def __del__(self):
        try:
            self.o.lib.orc_merkle_free(self.h)
        except Exception:
            pass


class OracleBatch:
    def __init__(self, o, cols, rate_bits, cap_height, from_values, threads):
        self.o = o
        c2 = u64(cols)
        self.ncols, self.n = c2.shape
        self.rate_bits, self.cap_height = rate_bits, cap_height
        self.h = _vp(o.lib.orc_batch_new(_p(c2), _sz(self.ncols), _sz(self.n), _u32(rate_bits), _u32(cap_height),
                                         1 if from_values else 0, _u32(threads)))

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.o.lib.orc_batch_cap(self.h, _p(out))
        return out

    @property
    def polynomials(self):
        out = np.empty((self.ncols, self.n), dtype=np.uint64)
        self.o.lib.orc_batch_coeffs(self.h, _p(out))
        return out

    def leaves(self):
        out = np.empty((self.n << self.rate_bits, self.ncols), dtype=np.uint64)
        self.o.lib.orc_batch_leaves(self.h, _p(out))
        return out

    def get_leaf(self, i):
        out = np.empty(self.ncols, dtype=np.uint64)
        self.o.lib.orc_batch_leaf(self.h, _sz(i), _p(out))
        return out

    def prove(self, i):
        out = np.empty((64, 4), dtype=np.uint64)
        n = self.o.lib.orc_batch_prove(self.h, _sz(i), _p(out))
        return out[:n].copy()

    def __del__(self):
        try:
            self.o.lib.orc_batch_free(self.h)
        except Exception:
            pass


class OracleCircuit:
    """build_matmul_circuit + witness + prove + verify of the CPU restatement (oracle/gl_circuit.hpp, gl_prover.hpp)."""
    INFO = ["degree_bits", "num_constants", "num_gate_constraints", "num_partial_products", "num_public_inputs",
            "num_selectors", "num_fri_rounds", "final_poly_len", "pi_row", "constant_row", "num_arith_ops", "num_poseidon_rows"]

    def __init__(self, o, m, threads=1, verifier_only=False):
        self.o, self.m = o, m
        lib = o.lib
        lib.orc_circuit_new_verifier_only.restype = _vp
        lib.orc_verify_bytes_message.restype = ctypes.c_char_p
        lib.orc_circuit_new.restype = _vp
        lib.orc_witness_new.restype = _vp
        lib.orc_witness_from_matrix.restype = _vp
        lib.orc_prove.restype = _vp
        lib.orc_proof_bytes.restype = _sz
        lib.orc_proof_challenges.restype = _sz
        lib.orc_proof_query_indices.restype = _sz
        lib.orc_witness_public_inputs.restype = _sz
        lib.orc_circuit_gate_order.restype = _sz
        lib.orc_verify_message.restype = ctypes.c_char_p
        self.h = _vp(lib.orc_circuit_new_verifier_only(_sz(m)) if verifier_only else lib.orc_circuit_new(_sz(m), _u32(threads)))
        info = np.zeros(12, dtype=np.uint64)
        lib.orc_circuit_info(self.h, _p(info))
        self.info = dict(zip(self.INFO, (int(x) for x in info)))
        self.n = 1 << self.info["degree_bits"]

    @classmethod
    def of_kind(cls, o, kind, param, threads=1):
        """gl_circuit.hpp build_test_circuit: kind 1 = hash-only (param public inputs), kind 2 = arithmetic chain, no PIs."""
        self = cls.__new__(cls)
        self.o, self.m = o, 0
        lib = o.lib
        for fn in ("orc_circuit_new_kind", "orc_witness_new", "orc_witness_from_matrix", "orc_prove"):
            getattr(lib, fn).restype = _vp
        for fn in ("orc_proof_bytes", "orc_proof_challenges", "orc_proof_query_indices", "orc_witness_public_inputs", "orc_circuit_gate_order",
                   "orc_circuit_selector_groups", "orc_circuit_num_inputs"):
            getattr(lib, fn).restype = _sz
        lib.orc_verify_message.restype = ctypes.c_char_p
        lib.orc_verify_bytes_message.restype = ctypes.c_char_p
        self.h = _vp(lib.orc_circuit_new_kind(ctypes.c_int(kind), _sz(param), _u32(threads)))
        info = np.zeros(12, dtype=np.uint64)
        lib.orc_circuit_info(self.h, _p(info))
        self.info = dict(zip(self.INFO, (int(x) for x in info)))
        self.n = 1 << self.info["degree_bits"]
        return self

    def data_bytes(self, kind):
        """kind 0: CommonCircuitData::to_bytes, 1: VerifierCircuitData::to_bytes, by the oracle's own writer."""
        lib = self.o.lib
        lib.orc_circuit_data_bytes.restype = _sz
        n = lib.orc_circuit_data_bytes(self.h, ctypes.c_int(kind), None, _sz(0))
        buf = np.empty(n, dtype=np.uint8)
        lib.orc_circuit_data_bytes(self.h, ctypes.c_int(kind), _p(buf), _sz(n))
        return buf.tobytes()

    def product_desc(self):
        """The gl_circuit_desc (plonky2_demo_amd._lib.CircuitDesc) of this circuit, for gl_circuit_create / gl_verify."""
        from plonky2_demo_amd._lib import CircuitDesc
        lib = self.o.lib
        lib.orc_circuit_selector_groups.restype = _sz
        d = CircuitDesc()
        i = self.info
        d.degree_bits, d.num_wires, d.num_routed_wires, d.num_constants = i["degree_bits"], 135, 80, i["num_constants"]
        d.num_selectors, d.num_challenges, d.quotient_degree_factor = i["num_selectors"], 2, 8
        d.rate_bits, d.cap_height, d.proof_of_work_bits, d.num_query_rounds = 3, 4, 16, 28
        d.num_fri_rounds = i["num_fri_rounds"]
        for r in range(d.num_fri_rounds):
            d.fri_arity_bits[r] = 4
        d.num_public_inputs = i["num_public_inputs"]
        order = self.gate_order()
        d.num_gates = len(order)
        groups = np.zeros(3 * 16, dtype=np.uint64)
        lib.orc_circuit_selector_groups(self.h, _p(groups))
        for g, t in enumerate(order):
            d.gate_types[g] = t
            d.gate_selector_index[g], d.gate_group_start[g], d.gate_group_end[g] = int(groups[3 * g]), int(groups[3 * g + 1]), int(groups[3 * g + 2])
        k = np.zeros(80, dtype=np.uint64)
        lib.orc_circuit_k_is(self.h, _p(k))
        for j in range(80):
            d.k_is[j] = int(k[j])
        li, rows = np.zeros(3, dtype=np.uint64), np.zeros(4 * 8, dtype=np.uint64)
        lut, gate_params = np.zeros(2048, dtype=np.uint16), np.zeros(16, dtype=np.uint8)
        lib.orc_circuit_lookup_info(self.h, _p(li), _p(rows), lut.ctypes.data_as(ctypes.c_void_p), gate_params.ctypes.data_as(ctypes.c_void_p))
        d.num_lookup_polys, d.num_lookup_selectors, d.num_luts = [int(x) for x in li]
        assert d.num_luts <= 4
        for t in range(d.num_luts):
            d.last_lu_row[t], d.last_lut_row[t], d.first_lut_row[t], d.lut_len[t] = [int(x) for x in rows[4 * t:4 * t + 4]]
        for j in range(2 * sum(d.lut_len[t] for t in range(d.num_luts))):
            d.lut[j] = int(lut[j])
        for g in range(d.num_gates):
            d.gate_params[g] = int(gate_params[g])
        return d

    @property
    def digest(self):
        out = np.empty(4, dtype=np.uint64)
        self.o.lib.orc_circuit_digest(self.h, _p(out))
        return out

    @property
    def constants_sigmas_cap(self):
        out = np.empty((16, 4), dtype=np.uint64)
        self.o.lib.orc_circuit_cs_cap(self.h, _p(out))
        return out

    def constants_sigmas(self):
        out = np.empty((self.info["num_constants"] + 80, self.n), dtype=np.uint64)
        self.o.lib.orc_circuit_constants_sigmas(self.h, _p(out))
        return out

    def row_gates(self):
        out = np.empty(self.n, dtype=np.uint8)
        self.o.lib.orc_circuit_row_gates(self.h, _p(out))
        return out

    def gate_order(self):
        out = np.empty(32, dtype=np.uint8)
        k = self.o.lib.orc_circuit_gate_order(self.h, _p(out))
        return out[:k].tolist()

    def witness(self, a, b, filler_seed=0x504C4F4E4B5932):
        return OracleWitness(self, u64(a).reshape(-1), u64(b).reshape(-1), filler_seed)

    def verify_bytes(self, proof_bytes, constants_sigmas_cap, circuit_digest):
        """The restated native verifier on raw ProofWithPublicInputs bytes + VerifierOnlyCircuitData."""
        buf = np.frombuffer(proof_bytes, dtype=np.uint8).copy()
        cap, dg = u64(constants_sigmas_cap), u64(circuit_digest)
        r = self.o.lib.orc_verify_bytes(self.h, _p(cap), _p(dg), _p(buf), _sz(buf.size))
        return r == 0, self.o.lib.orc_verify_bytes_message().decode()

    def __del__(self):
        try:
            self.o.lib.orc_circuit_free(self.h)
        except Exception:
            pass


class OracleWitness:
    def __init__(self, circuit, a, b, seed):
        self.c = circuit
        lib = circuit.o.lib
        lib.orc_circuit_num_inputs.restype = _sz
        nb = ctypes.c_size_t()
        na = lib.orc_circuit_num_inputs(circuit.h, ctypes.byref(nb))
        assert a.size == na and b.size == nb.value, (a.size, b.size, na, nb.value)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint64)           # a valid pointer for circuits without second-operand targets
        self.h = _vp(lib.orc_witness_new(circuit.h, _p(a), _p(b), _u64(seed)))

    def wires(self):
        out = np.empty((135, self.c.n), dtype=np.uint64)
        self.c.o.lib.orc_witness_wires(self.h, _p(out))
        return out

    def public_inputs(self):
        k = self.c.o.lib.orc_witness_public_inputs(self.h, None)
        out = np.empty(k, dtype=np.uint64)
        self.c.o.lib.orc_witness_public_inputs(self.h, _p(out))
        return out

    def prove(self, threads=1):
        h = self.c.o.lib.orc_prove(self.c.h, self.h, _u32(threads))
        if not h:
            raise RuntimeError("oracle prover failed (quotient not divisible or zeta in H)")
        return OracleProof(self.c, _vp(h))

    def __del__(self):
        try:
            self.c.o.lib.orc_witness_free(self.h)
        except Exception:
            pass


class OracleProof:
    def __init__(self, circuit, h):
        self.c, self.h = circuit, h

    def to_bytes(self):
        lib = self.c.o.lib
        k = lib.orc_proof_bytes(self.h, None, _sz(0))
        buf = np.empty(k, dtype=np.uint8)
        lib.orc_proof_bytes(self.h, _p(buf), _sz(k))
        return buf.tobytes()

    def challenges(self):
        out = np.zeros(64, dtype=np.uint64)
        k = self.c.o.lib.orc_proof_challenges(self.h, _p(out))
        v = [int(x) for x in out[:k]]
        return {"betas": v[0:2], "gammas": v[2:4], "alphas": v[4:6], "zeta": v[6:8], "fri_alpha": v[8:10], "pow_witness": v[10],
                "public_inputs_hash": v[11:15], "fri_betas": [v[i:i + 2] for i in range(15, k, 2)]}

    def caps(self):
        out = np.empty((3, 16, 4), dtype=np.uint64)
        self.c.o.lib.orc_proof_caps(self.h, _p(out))
        return out

    def zs_partial_products(self, ncols=20):
        out = np.empty((ncols, self.c.n), dtype=np.uint64)
        self.c.o.lib.orc_proof_zs_partial_products(self.h, _p(out))
        return out

    def quotient_chunks(self):
        out = np.empty((16, self.c.n), dtype=np.uint64)
        self.c.o.lib.orc_proof_quotient_chunks(self.h, _p(out))
        return out

    def final_poly_initial(self):
        out = np.empty((self.c.n, 2), dtype=np.uint64)
        self.c.o.lib.orc_proof_final_poly_initial(self.h, _p(out))
        return out

    def query_indices(self):
        out = np.zeros(64, dtype=np.uint64)
        k = self.c.o.lib.orc_proof_query_indices(self.h, _p(out))
        return [int(x) for x in out[:k]]

    def verify(self):
        r = self.c.o.lib.orc_verify(self.c.h, self.h)
        return r == 0, self.c.o.lib.orc_verify_message().decode()

    def tamper(self, what):
        self.c.o.lib.orc_proof_tamper(self.h, what)

    def __del__(self):
        try:
            self.c.o.lib.orc_proof_free(self.h)
        except Exception:
            pass


_cached = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build()
        _cached = Oracle(ctypes.CDLL(so))
    return _cached


def splitmix64(seed, count):
    """Deterministic canonical field elements (splitmix64 streams mod p)."""
    out = np.empty(count, dtype=np.uint64)
    x = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(count):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z = z ^ (z >> 31)
        out[i] = z % P
    return out


def rand_field(seed, shape):
    """Fast seeded canonical field elements for large test inputs (numpy PCG64, rejection-free mod p)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, P, size=shape, dtype=np.uint64, endpoint=False)
    return a
