"""Host-side mirror of the reference's operator interfaces for the hot path (numpy in / numpy out).

Everything here is plumbing over the C ABI: no arithmetic happens in Python.  Arrays are uint64, an
extension element is a pair, a digest is four words.  Reference lines are cited per function.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib

GOLDILOCKS_ORDER = 0xFFFFFFFF00000001   # field/src/goldilocks_field.rs:152
COSET_SHIFT = 7                          # field/src/types.rs:437-439, goldilocks_field.rs:80


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _log2_strict(n):
    lg = int(n).bit_length() - 1
    if n <= 0 or (1 << lg) != n:
        raise ValueError("length %d is not a power of two" % n)   # util/src/lib.rs:35-40 panics
    return lg


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib.gl_dev_alloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = _u64(arr)
        assert arr.nbytes <= self.nbytes
        check(lib.gl_copy_h2d(self.ctx.handle, self.ptr, _p(arr), arr.nbytes))
        return self

    def download(self, shape):
        out = np.empty(shape, dtype=np.uint64)
        assert out.nbytes <= self.nbytes
        check(lib.gl_copy_d2h(self.ctx.handle, _p(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib.gl_dev_free(self.ctx.handle, self.ptr))      # a closed context (handle None) is accepted
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per (device, stream).  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream, so that torch events bracket the library's kernels."""

    def __init__(self, device=0, stream=None):
        h = ctypes.c_void_p()
        check(lib.gl_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h)))
        self.handle = h.value
        self.device = device

    def synchronize(self):
        check(lib.gl_ctx_synchronize(self.handle))

    def set_scratch_elems(self, elems):
        check(lib.gl_ctx_set_scratch_elems(self.handle, int(elems)))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def capture_intermediates(self, on=True):
        """Keep each proof's Z / partial-product values and quotient chunks on the host (parity tests)."""
        check(lib.gl_ctx_capture_intermediates(self.handle, 1 if on else 0))

    def timing(self, on=True):
        check(lib.gl_ctx_timing_reset(self.handle))
        check(lib.gl_ctx_timing_enable(self.handle, 1 if on else 0))

    def timing_report(self):
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        check(lib.gl_ctx_timing_report(self.handle, buf, len(buf)))
        return json.loads(buf.value.decode())

    def close(self):
        """Drops this object's reference to the context (gl_ctx_destroy).  Batches / trees / circuits created on it hold
        references of their own, so they stay valid and may be freed afterwards in any order; buffers from alloc() must be
        freed before."""
        if self.handle:
            lib.gl_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # safe in any finalisation order: the library tears the context down when its last handle is freed
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context(0)
    return _default


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


# ---------------------------------------------------------------------------------------------- field
def field_op(op, a, b=None, c=None, ctx=None):
    """op: 0 add 1 sub 2 mul 3 neg 4 inverse 5 canonicalise 6 a+b*c 7 a*2^(b%192)."""
    ctx = _ctx(ctx)
    a = _u64(a)
    n = a.size
    da = ctx.alloc(max(8, a.nbytes)).upload(a)
    db = ctx.alloc(max(8, a.nbytes)).upload(_u64(b)) if b is not None else None
    dc = ctx.alloc(max(8, a.nbytes)).upload(_u64(c)) if c is not None else None
    do = ctx.alloc(max(8, a.nbytes))
    check(lib.gl_field_op(ctx.handle, op, da.ptr, db.ptr if db else None, dc.ptr if dc else None, do.ptr, n))
    return do.download(a.shape)


def ext_op(op, a, b=None, ctx=None):
    ctx = _ctx(ctx)
    a = _u64(a)
    n = a.size // 2
    da = ctx.alloc(max(8, a.nbytes)).upload(a)
    db = ctx.alloc(max(8, a.nbytes)).upload(_u64(b)) if b is not None else None
    do = ctx.alloc(max(8, a.nbytes))
    check(lib.gl_ext_op(ctx.handle, op, da.ptr, db.ptr if db else None, do.ptr, n))
    return do.download(a.shape)


# ------------------------------------------------------------------------------------------------ NTT
def _transform(kind, arr, shift=None, ctx=None):
    ctx = _ctx(ctx)
    arr = _u64(arr)
    single = arr.ndim == 1
    a2 = arr.reshape(1, -1) if single else arr
    batch, n = a2.shape
    lg = _log2_strict(n)
    d = ctx.alloc(max(8, a2.nbytes)).upload(a2)
    if kind == "fft":
        check(lib.gl_ntt_forward(ctx.handle, d.ptr, lg, batch))
    elif kind == "ifft":
        check(lib.gl_ntt_inverse(ctx.handle, d.ptr, lg, batch))
    elif kind == "coset_fft":
        check(lib.gl_ntt_coset_forward(ctx.handle, d.ptr, lg, batch, shift))
    elif kind == "coset_ifft":
        check(lib.gl_ntt_coset_inverse(ctx.handle, d.ptr, lg, batch, shift))
    out = d.download(a2.shape)
    return out.reshape(-1) if single else out


def fft(coeffs, ctx=None):
    """field::fft::fft (field/src/fft.rs:52-65): values[i] = P(w^i); [n] or [batch][n]."""
    return _transform("fft", coeffs, ctx=ctx)


def ifft(values, ctx=None):
    """field::fft::ifft (field/src/fft.rs:67-95)."""
    return _transform("ifft", values, ctx=ctx)


def coset_fft(coeffs, shift=COSET_SHIFT, ctx=None):
    """PolynomialCoeffs::coset_fft (field/src/polynomial/mod.rs:276-295)."""
    return _transform("coset_fft", coeffs, shift=shift, ctx=ctx)


def coset_ifft(values, shift=COSET_SHIFT, ctx=None):
    """PolynomialValues::coset_ifft (field/src/polynomial/mod.rs:58-70)."""
    return _transform("coset_ifft", values, shift=shift, ctx=ctx)


def lde_onto_coset(coeffs, rate_bits, ctx=None):
    """coeffs.lde(rate_bits).coset_fft_with_options(7, Some(rate_bits)) (plonky2/src/fri/oracle.rs:111-118)."""
    ctx = _ctx(ctx)
    c2 = _u64(coeffs)
    single = c2.ndim == 1
    if single:
        c2 = c2.reshape(1, -1)
    batch, n = c2.shape
    lg = _log2_strict(n)
    src = ctx.alloc(max(8, c2.nbytes)).upload(c2)
    dst = ctx.alloc(c2.nbytes << rate_bits)
    check(lib.gl_ntt_coset_lde(ctx.handle, src.ptr, lg, rate_bits, batch, dst.ptr))
    out = dst.download((batch, n << rate_bits))
    return out.reshape(-1) if single else out


# ------------------------------------------------------------------------------------------- hashing
def poseidon(states, ctx=None):
    """Poseidon::poseidon (plonky2/src/hash/poseidon.rs:598-609) on [12] or [count][12]."""
    ctx = _ctx(ctx)
    s = _u64(states)
    if s.shape[-1] != 12:
        raise ValueError("Poseidon state width is 12")
    d = ctx.alloc(s.nbytes).upload(s)
    check(lib.gl_poseidon_permute(ctx.handle, d.ptr, s.size // 12))
    return d.download(s.shape)


def hash_or_noop(rows, ctx=None):
    """Hasher::hash_or_noop (plonky2/src/plonk/config.rs:55-66) per row of [count][len] -> [count][4]."""
    ctx = _ctx(ctx)
    r = _u64(rows)
    single = r.ndim == 1
    if single:
        r = r.reshape(1, -1)
    count, ln = r.shape
    if ln == 0:
        return np.zeros((4,) if single else (count, 4), dtype=np.uint64)
    d = ctx.alloc(max(8, r.nbytes)).upload(r)
    o = ctx.alloc(max(32, count * 32))
    check(lib.gl_hash_rows(ctx.handle, d.ptr, count, ln, o.ptr))
    out = o.download((count, 4))
    return out[0] if single else out


class MerkleTree:
    """plonky2::hash::merkle_tree::MerkleTree (merkle_tree.rs:39-207) with device-resident digests."""

    def __init__(self, leaves, cap_height, ctx=None):
        self.ctx = _ctx(ctx)
        l2 = _u64(leaves)
        if l2.ndim != 2:
            raise ValueError("leaves must be [num_leaves][leaf_len]")
        self.num_leaves, self.leaf_len = l2.shape
        self.cap_height = cap_height
        self.leaves = l2
        h = ctypes.c_void_p()
        check(lib.gl_merkle_new(self.ctx.handle, _p(l2), self.num_leaves, self.leaf_len, cap_height, ctypes.byref(h)))
        self.handle = h.value

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        check(lib.gl_merkle_cap(self.handle, _p(out)))
        return out

    def get(self, i):
        return self.leaves[i]

    def prove(self, leaf_index):
        n = ctypes.c_uint32()
        out = np.empty((64, 4), dtype=np.uint64)
        check(lib.gl_merkle_prove(self.handle, leaf_index, _p(out), ctypes.byref(n)))
        return out[: n.value].copy()

    def __del__(self):
        try:
            if self.handle:
                lib.gl_merkle_free(self.handle)
                self.handle = None
        except Exception:
            pass


class PolynomialBatch:
    """plonky2::fri::oracle::PolynomialBatch (fri/oracle.rs:30-133), device-resident."""

    def __init__(self, handle, ctx, rate_bits, cap_height):
        self.handle, self.ctx, self.rate_bits, self.cap_height = handle, ctx, rate_bits, cap_height
        self.ncols = lib.gl_batch_ncols(handle)
        self.degree = lib.gl_batch_degree(handle)
        self.degree_log = _log2_strict(self.degree)

    @classmethod
    def _from_host(cls, fn, cols, rate_bits, blinding, cap_height, ctx):
        ctx = _ctx(ctx)
        cols = [_u64(c) for c in cols]
        if not cols:
            raise ValueError("empty batch")
        n = cols[0].size
        if any(c.size != n for c in cols):
            raise ValueError("Polynomial degrees inconsistent")   # oracle.rs:114
        ptrs = (ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        h = ctypes.c_void_p()
        check(fn(ctx.handle, ptrs, len(cols), n, rate_bits, 1 if blinding else 0, cap_height, ctypes.byref(h)))
        return cls(h.value, ctx, rate_bits, cap_height)

    @classmethod
    def from_values(cls, values, rate_bits, blinding, cap_height, ctx=None):
        """PolynomialBatch::from_values (fri/oracle.rs:43-66)."""
        return cls._from_host(lib.gl_batch_from_values, values, rate_bits, blinding, cap_height, ctx)

    @classmethod
    def from_coeffs(cls, polynomials, rate_bits, blinding, cap_height, ctx=None):
        """PolynomialBatch::from_coeffs (fri/oracle.rs:68-98)."""
        return cls._from_host(lib.gl_batch_from_coeffs, polynomials, rate_bits, blinding, cap_height, ctx)

    @classmethod
    def from_device(cls, d_ptr, ncols, n, rate_bits, cap_height, is_values, ctx=None):
        ctx = _ctx(ctx)
        h = ctypes.c_void_p()
        check(lib.gl_batch_from_device(ctx.handle, d_ptr, ncols, n, rate_bits, cap_height, 1 if is_values else 0, ctypes.byref(h)))
        return cls(h.value, ctx, rate_bits, cap_height)

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        check(lib.gl_batch_cap(self.handle, _p(out)))
        return out

    @property
    def polynomials(self):
        out = np.empty((self.ncols, self.degree), dtype=np.uint64)
        check(lib.gl_batch_coeffs(self.handle, _p(out)))
        return out

    def lde_values(self):
        out = np.empty((self.ncols, self.degree << self.rate_bits), dtype=np.uint64)
        check(lib.gl_batch_lde(self.handle, _p(out)))
        return out

    def get_leaf(self, i):
        out = np.empty(self.ncols, dtype=np.uint64)
        check(lib.gl_batch_get_leaf(self.handle, i, _p(out)))
        return out

    def get_lde_values(self, index, step):
        """PolynomialBatch::get_lde_values (fri/oracle.rs:128-133)."""
        out = np.empty(self.ncols, dtype=np.uint64)
        check(lib.gl_batch_get_lde_values(self.handle, index, step, _p(out)))
        return out

    def prove(self, leaf_index):
        n = ctypes.c_uint32()
        out = np.empty((64, 4), dtype=np.uint64)
        check(lib.gl_batch_prove(self.handle, leaf_index, _p(out), ctypes.byref(n)))
        return out[: n.value].copy()

    handle_owned = True     # False for batches borrowed from a circuit

    def open_at(self, z, first_col=0, num_cols=None, ctx=None):
        """eval_commitment of OpeningSet::new (plonk/proof.rs:306-344): [num_cols][2] extension values."""
        num_cols = self.ncols - first_col if num_cols is None else num_cols
        out = np.empty((num_cols, 2), dtype=np.uint64)
        check(lib.gl_open_at((ctx or self.ctx).handle, self.handle, _p(_u64(z)), first_col, num_cols, _p(out)))
        return out

    def free(self):
        if self.handle:
            if self.handle_owned:
                lib.gl_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------- circuit and prove()
class MatmulCircuit:
    """Host side of the demo (plonky2/src/bin/matrix_mul.rs:25-67 + CircuitBuilder::build()): needs no GPU."""

    def __init__(self, m):
        h = ctypes.c_void_p()
        check(lib.gl_matmul_circuit_build(int(m), ctypes.byref(h)))
        self.handle, self.m = h.value, int(m)
        self.desc = _lib.CircuitDesc()
        check(lib.gl_host_circuit_desc(self.handle, ctypes.byref(self.desc)))
        self.degree_bits = self.desc.degree_bits
        self.n = 1 << self.degree_bits

    def row_gates(self):
        out = np.empty(self.n, dtype=np.uint8)
        check(lib.gl_host_circuit_row_gates(self.handle, _p(out)))
        return out

    def constants_sigmas(self):
        out = np.empty((self.desc.num_constants + 80, self.n), dtype=np.uint64)
        check(lib.gl_host_circuit_constants_sigmas(self.handle, _p(out)))
        return out

    def witness(self, a, b, filler_seed=0x504C4F4E4B5932):
        """(wires[135][n], public_inputs[3 m^2]) -- generate_partial_witness + full_witness (prover.rs:118-133)."""
        a, b = _u64(a).reshape(-1), _u64(b).reshape(-1)
        if a.size != self.m ** 2 or b.size != self.m ** 2:
            raise ValueError("a and b must be m x m")
        wires = np.empty((135, self.n), dtype=np.uint64)
        pis = np.empty(3 * self.m ** 2, dtype=np.uint64)
        check(lib.gl_matmul_witness(self.handle, _p(a), _p(b), filler_seed, _p(wires), _p(pis)))
        return wires, pis

    def verify(self, proof_bytes, constants_sigmas_cap, circuit_digest):
        """VerifierCircuitData::verify (plonk/circuit_data.rs:208-215): (accepted, reason).  Host code, no GPU needed."""
        buf = np.frombuffer(bytes(proof_bytes), dtype=np.uint8)
        cap, dig = _u64(constants_sigmas_cap), _u64(circuit_digest)
        if cap.size != 4 << self.desc.cap_height or dig.size != 4:
            raise ValueError("cap must be [2^cap_height][4], digest [4]")
        st = lib.gl_host_circuit_verify(self.handle, _p(cap), _p(dig), _p(buf), buf.size)
        if st == _lib.GL_OK:
            return True, ""
        if st == _lib.GL_ERR_VERIFY:
            return False, (lib.gl_last_error() or b"").decode()
        check(st)

    def witness_generator(self, ctx=None):
        """Witness generation straight into HBM (GPU arithmetic rows + host hash-sponge rows), one per context."""
        return WitnessGenerator(self, _ctx(ctx))

    def build(self, ctx=None):
        """CircuitBuilder::build(): the device half (constants/sigmas commitment, digest)."""
        return CircuitData(self, _ctx(ctx))

    def __del__(self):
        try:
            if self.handle:
                lib.gl_host_circuit_free(self.handle)
                self.handle = None
        except Exception:
            pass


class WitnessGenerator:
    """generate_partial_witness + full_witness (plonk/prover.rs:118-133) for the matmul family, wire matrix in HBM."""

    def __init__(self, host, ctx):
        self.host, self.ctx = host, ctx
        h = ctypes.c_void_p()
        check(lib.gl_matmul_witgen_create(ctx.handle, host.handle, ctypes.byref(h)))
        self.handle = h.value

    def run(self, a, b, d_wires_ptr, filler_seed=0x504C4F4E4B5932):
        """Overwrites the device matrix d_wires[135][n]; returns the public inputs."""
        a, b = _u64(a).reshape(-1), _u64(b).reshape(-1)
        if a.size != self.host.m ** 2 or b.size != self.host.m ** 2:
            raise ValueError("a and b must be m x m")
        pis = np.empty(3 * self.host.m ** 2, dtype=np.uint64)
        self.public_inputs_hash = np.empty(4, dtype=np.uint64)      # by-product of the sponge rows
        check(lib.gl_matmul_witgen_run(self.handle, _p(a), _p(b), filler_seed, d_wires_ptr, _p(pis), _p(self.public_inputs_hash)))
        return pis

    def __del__(self):
        try:
            if self.handle:
                lib.gl_matmul_witgen_free(self.handle)
                self.handle = None
        except Exception:
            pass


def _prove_device(ctx, circuit_handle, n, d_wires_ptr, public_inputs, public_inputs_hash):
    pis = _u64(public_inputs)
    h = ctypes.c_void_p()
    if public_inputs_hash is None:
        check(lib.gl_prove_device(ctx.handle, circuit_handle, d_wires_ptr, _p(pis), pis.size, ctypes.byref(h)))
    else:
        check(lib.gl_prove_device_hashed(ctx.handle, circuit_handle, d_wires_ptr, _p(pis), pis.size, _p(_u64(public_inputs_hash)), ctypes.byref(h)))
    return Proof(h.value, n)


class HostColumns:
    """135 host vectors of n field elements with their pointer table (what a Rust caller passes as `*const *const u64`); built
    once so that a timed loop measures gl_prove_columns, not numpy."""

    def __init__(self, columns, n):
        self.cols = [np.ascontiguousarray(_u64(c)) for c in columns]
        if len(self.cols) != 135 or any(c.size != n for c in self.cols):
            raise ValueError("need 135 columns of n values")
        self.ptrs = (ctypes.c_void_p * 135)(*[c.ctypes.data for c in self.cols])


def _prove_columns(ctx, circuit_handle, n, columns, public_inputs):
    hc = columns if isinstance(columns, HostColumns) else HostColumns(columns, n)
    pis = _u64(public_inputs)
    h = ctypes.c_void_p()
    check(lib.gl_prove_columns(ctx.handle, circuit_handle, hc.ptrs, _p(pis), pis.size, ctypes.byref(h)))
    return Proof(h.value, n)


def _warm_up(ctx, handle):
    check(lib.gl_circuit_warm_up(ctx.handle, handle))


class _PhaseApi:
    """The phase-level seam of a device-resident circuit, for a caller that owns the Challenger; `self.desc` is the
    gl_circuit_desc, `self.handle` the gl_circuit.  Circuits with the lookup argument pass the 8 delta challenges."""

    def warm_up(self, ctx=None):
        """gl_circuit_warm_up: one throw-away pass of the proving pipeline on `ctx` (default: the circuit's context), so that the first
        proof does not pay the one-time costs (kernel code objects, twiddle tables, pool growth)."""
        _warm_up(_ctx(ctx) if ctx is not None else self.ctx, self.handle)
        return self

    def _batch(self, handle, ctx):
        return PolynomialBatch(handle, ctx, self.desc.rate_bits, self.desc.cap_height)

    @property
    def constants_sigmas_batch(self):
        """`prover_data.constants_sigmas_commitment` (borrowed: owned by the circuit)."""
        b = self._batch(lib.gl_circuit_constants_sigmas_batch(self.handle), self.ctx)
        b.handle_owned = False
        return b

    def partial_products(self, d_wires_ptr, betas, gammas, ctx=None, deltas=None):
        """all_wires_permutation_partial_products (+ compute_all_lookup_polys) + commitment (plonk/prover.rs:189-223)."""
        ctx = ctx or self.ctx
        h = ctypes.c_void_p()
        if deltas is None:
            check(lib.gl_partial_products(ctx.handle, self.handle, d_wires_ptr, _p(_u64(betas)), _p(_u64(gammas)), ctypes.byref(h)))
        else:
            check(lib.gl_partial_products_lookups(ctx.handle, self.handle, d_wires_ptr, _p(_u64(betas)), _p(_u64(gammas)), _p(_u64(deltas)), ctypes.byref(h)))
        return self._batch(h.value, ctx)

    def quotient_polys(self, wires_batch, zs_batch, public_inputs_hash, betas, gammas, alphas, ctx=None, deltas=None):
        """compute_quotient_polys + chunking + commitment (plonk/prover.rs:229-271)."""
        ctx = ctx or self.ctx
        h = ctypes.c_void_p()
        if deltas is None:
            check(lib.gl_quotient_polys(ctx.handle, self.handle, wires_batch.handle, zs_batch.handle, _p(_u64(public_inputs_hash)),
                                        _p(_u64(betas)), _p(_u64(gammas)), _p(_u64(alphas)), ctypes.byref(h)))
        else:
            check(lib.gl_quotient_polys_lookups(ctx.handle, self.handle, wires_batch.handle, zs_batch.handle, _p(_u64(public_inputs_hash)),
                                                _p(_u64(betas)), _p(_u64(gammas)), _p(_u64(alphas)), _p(_u64(deltas)), ctypes.byref(h)))
        return self._batch(h.value, ctx)

    def fri(self, batches, zeta, alpha, ctx=None):
        """PolynomialBatch::prove_openings up to fri_proof (fri/oracle.rs:162-204)."""
        return FriProver(self, batches, zeta, alpha, ctx or self.ctx)


class CircuitData(_PhaseApi):
    """plonky2::plonk::circuit_data::CircuitData for the prover: `prove(wires, public_inputs)` mirrors
    CircuitData::prove (circuit_data.rs:144-151) at the full-witness boundary."""

    def __init__(self, host, ctx):
        self.host, self.ctx, self.desc = host, ctx, host.desc
        h = ctypes.c_void_p()
        check(lib.gl_circuit_from_host(ctx.handle, host.handle, ctypes.byref(h)))
        self.handle = h.value

    @property
    def circuit_digest(self):
        out = np.empty(4, dtype=np.uint64)
        check(lib.gl_circuit_digest(self.handle, _p(out)))
        return out

    @property
    def constants_sigmas_cap(self):
        out = np.empty((1 << self.host.desc.cap_height, 4), dtype=np.uint64)
        check(lib.gl_circuit_constants_sigmas_cap(self.handle, _p(out)))
        return out

    def verify(self, proof):
        """CircuitData::verify (plonk/circuit_data.rs:153-155); `proof` is a Proof or its bytes."""
        by = proof.to_bytes() if hasattr(proof, "to_bytes") else proof
        return self.host.verify(by, self.constants_sigmas_cap, self.circuit_digest)

    def prove(self, wires, public_inputs):
        wires, pis = _u64(wires), _u64(public_inputs)
        if wires.shape != (135, self.host.n):
            raise ValueError("wire matrix must be [135][n]")
        h = ctypes.c_void_p()
        check(lib.gl_prove(self.ctx.handle, self.handle, _p(wires), _p(pis), pis.size, ctypes.byref(h)))
        return Proof(h.value, self.host.n)

    def prove_columns(self, columns, public_inputs):
        """prove() from one host array per wire, as the reference keeps MatrixWitness.wire_values (iop/witness.rs:256-258):
        the drop-in entry INTEGRATION.md patches into plonk/prover.rs:145.  `columns` may be a HostColumns (pointer table built once)."""
        return _prove_columns(self.ctx, self.handle, self.host.n, columns, public_inputs)

    def prove_device(self, d_wires_ptr, public_inputs, public_inputs_hash=None):
        """prove() with the witness matrix already in HBM (raw device pointer to [135][n] u64)."""
        return _prove_device(self.ctx, self.handle, self.host.n, d_wires_ptr, public_inputs, public_inputs_hash)

    def __del__(self):
        try:
            if self.handle:
                lib.gl_circuit_free(self.handle)
                self.handle = None
        except Exception:
            pass


class GenericCircuitData(_PhaseApi):
    """Prover + verifier for ANY circuit over the demo's gate set, given what CircuitBuilder::build() produces: the descriptor
    (CommonCircuitData) and the constants || sigmas value columns (gl_circuit_create)."""

    def __init__(self, desc, constants_sigmas, ctx=None):
        self.ctx, self.desc = _ctx(ctx), desc
        self.n = 1 << desc.degree_bits
        cs = _u64(constants_sigmas)
        if cs.shape != (desc.num_constants + 80, self.n):
            raise ValueError("constants_sigmas must be [num_constants + 80][n]")
        h = ctypes.c_void_p()
        check(lib.gl_circuit_create(self.ctx.handle, ctypes.byref(desc), _p(cs), ctypes.byref(h)))
        self.handle = h.value
        self.desc = type(desc)()                      # the completed description (lookup rows read from the selector columns)
        check(lib.gl_circuit_description(self.handle, ctypes.byref(self.desc)))

    @property
    def circuit_digest(self):
        out = np.empty(4, dtype=np.uint64)
        check(lib.gl_circuit_digest(self.handle, _p(out)))
        return out

    @property
    def constants_sigmas_cap(self):
        out = np.empty((1 << self.desc.cap_height, 4), dtype=np.uint64)
        check(lib.gl_circuit_constants_sigmas_cap(self.handle, _p(out)))
        return out

    def prove(self, wires, public_inputs):
        wires, pis = _u64(wires), _u64(public_inputs)
        if wires.shape != (135, self.n):
            raise ValueError("wire matrix must be [135][n]")
        if pis.size == 0:
            pis = np.zeros(1, dtype=np.uint64)[:0]
        h = ctypes.c_void_p()
        keep = np.zeros(1, dtype=np.uint64) if pis.size == 0 else pis          # a valid pointer even for zero public inputs
        check(lib.gl_prove(self.ctx.handle, self.handle, _p(wires), _p(keep), pis.size, ctypes.byref(h)))
        return Proof(h.value, self.n)

    def verify(self, proof):
        by = proof.to_bytes() if hasattr(proof, "to_bytes") else proof
        buf = np.frombuffer(bytes(by), dtype=np.uint8)
        st = lib.gl_verify(ctypes.byref(self.desc), _p(self.constants_sigmas_cap), _p(self.circuit_digest), _p(buf), buf.size)
        if st == _lib.GL_OK:
            return True, ""
        if st == _lib.GL_ERR_VERIFY:
            return False, (lib.gl_last_error() or b"").decode()
        check(st)

    def __del__(self):
        try:
            if self.handle:
                lib.gl_circuit_free(self.handle)
                self.handle = None
        except Exception:
            pass


# ------------------------------------------------------------------------------- circuit data as bytes (host code)
def common_data_to_bytes(desc):
    """CommonCircuitData::to_bytes of a gl_circuit_desc (util/serialization/mod.rs:1736-1790)."""
    n = ctypes.c_size_t()
    check(lib.gl_common_data_to_bytes(ctypes.byref(desc), None, 0, ctypes.byref(n)))
    buf = np.empty(n.value, dtype=np.uint8)
    check(lib.gl_common_data_to_bytes(ctypes.byref(desc), _p(buf), buf.size, ctypes.byref(n)))
    return buf.tobytes()


def common_data_from_bytes(data):
    """-> (CircuitDesc, bytes consumed); GL_ERR_UNSUPPORTED for gates / features outside the demo's set."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    d, used = _lib.CircuitDesc(), ctypes.c_size_t()
    check(lib.gl_common_data_from_bytes(_p(buf), buf.size, ctypes.byref(d), ctypes.byref(used)))
    return d, used.value


def verifier_only_to_bytes(constants_sigmas_cap, circuit_digest):
    cap, dig = _u64(constants_sigmas_cap).reshape(-1, 4), _u64(circuit_digest)
    h = _log2_strict(cap.shape[0])
    n = ctypes.c_size_t()
    buf = np.empty(8 + 32 * cap.shape[0] + 32, dtype=np.uint8)
    check(lib.gl_verifier_only_to_bytes(h, _p(cap), _p(dig), _p(buf), buf.size, ctypes.byref(n)))
    return buf[: n.value].tobytes()


def verifier_only_from_bytes(data):
    """-> (cap[2^h][4], digest[4], bytes consumed)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    h, used = ctypes.c_uint32(), ctypes.c_size_t()
    dig = np.empty(4, dtype=np.uint64)
    check(lib.gl_verifier_only_from_bytes(_p(buf), buf.size, ctypes.byref(h), None, 0, _p(dig), ctypes.byref(used)))
    cap = np.empty((1 << h.value, 4), dtype=np.uint64)
    check(lib.gl_verifier_only_from_bytes(_p(buf), buf.size, ctypes.byref(h), _p(cap), cap.size, _p(dig), ctypes.byref(used)))
    return cap, dig, used.value


def verifier_data_to_bytes(desc, constants_sigmas_cap, circuit_digest):
    """VerifierCircuitData::to_bytes = verifier_only || common (util/serialization/mod.rs:1908-1919)."""
    return verifier_only_to_bytes(constants_sigmas_cap, circuit_digest) + common_data_to_bytes(desc)


def verify_bytes(verifier_data, proof_bytes):
    """VerifierCircuitData::from_bytes(verifier_data).verify(proof): (accepted, reason)."""
    vd = np.frombuffer(bytes(verifier_data), dtype=np.uint8)
    pb = np.frombuffer(bytes(proof_bytes), dtype=np.uint8)
    st = lib.gl_verify_bytes(_p(vd), vd.size, _p(pb), pb.size)
    if st == _lib.GL_OK:
        return True, ""
    if st == _lib.GL_ERR_VERIFY:
        return False, (lib.gl_last_error() or b"").decode()
    check(st)


class GenericProverPool:
    """gl_prover_pool_create_generic / _prove_columns: `lanes` warmed-up contexts and host threads inside the library for ANY circuit
    (description + constants || sigmas); a batch of host witnesses (135 column vectors each) becomes one call."""

    def __init__(self, desc, constants_sigmas, lanes=4, device=0):
        self.desc, self.n = desc, 1 << desc.degree_bits
        cs = _u64(constants_sigmas)
        if cs.shape != (desc.num_constants + 80, self.n):
            raise ValueError("constants_sigmas must be [num_constants + 80][n]")
        h = ctypes.c_void_p()
        check(lib.gl_prover_pool_create_generic(device, ctypes.byref(desc), _p(cs), lanes, ctypes.byref(h)))
        self.handle = h.value

    def prove_columns(self, witnesses):
        """witnesses: list of (columns, public_inputs), columns = 135 arrays of n u64 (or a [135][n] matrix); returns the Proofs in order."""
        k = len(witnesses)
        keep, col_ptrs, pi_ptrs = [], (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
        for i, (cols, pis) in enumerate(witnesses):
            cols = [np.ascontiguousarray(_u64(c).reshape(-1)) for c in cols]
            if len(cols) != 135 or any(c.size != self.n for c in cols):
                raise ValueError("a witness is 135 columns of n values")
            arr = (ctypes.c_void_p * 135)(*[c.ctypes.data for c in cols])
            pv = np.ascontiguousarray(_u64(pis).reshape(-1))
            if pv.size != self.desc.num_public_inputs:
                raise ValueError("wrong number of public inputs")
            keep.append((cols, arr, pv))
            col_ptrs[i] = ctypes.addressof(arr)
            pi_ptrs[i] = pv.ctypes.data if pv.size else None
        out = (ctypes.c_void_p * k)()
        st = lib.gl_prover_pool_prove_columns(self.handle, k, col_ptrs, pi_ptrs, out)
        proofs = [Proof(out[i], self.n) if out[i] else None for i in range(k)]
        check(st)
        return proofs

    def close(self):
        if self.handle:
            lib.gl_prover_pool_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProverPool:
    """Many proofs in flight on one GPU from one call (gl_prover_pool_*): one circuit, `lanes` streams and host threads in C++."""

    def __init__(self, host, lanes=4, device=0):
        self.host = host
        h = ctypes.c_void_p()
        check(lib.gl_prover_pool_create(device, host.handle, lanes, ctypes.byref(h)))
        self.handle = h.value

    @property
    def circuit_digest(self):
        out = np.empty(4, dtype=np.uint64)
        check(lib.gl_circuit_digest(lib.gl_prover_pool_circuit(self.handle), _p(out)))
        return out

    @property
    def constants_sigmas_cap(self):
        out = np.empty((1 << self.host.desc.cap_height, 4), dtype=np.uint64)
        check(lib.gl_circuit_constants_sigmas_cap(lib.gl_prover_pool_circuit(self.handle), _p(out)))
        return out

    def prove_matmul(self, operands, filler_seeds=None):
        """operands: list of (a, b) m x m arrays; returns the list of Proof objects in the same order."""
        m2 = self.host.m ** 2
        aa = [np.ascontiguousarray(_u64(a).reshape(-1)) for a, _ in operands]
        bb = [np.ascontiguousarray(_u64(b).reshape(-1)) for _, b in operands]
        if any(x.size != m2 for x in aa + bb):
            raise ValueError("operands must be m x m")
        k = len(operands)
        pa = (ctypes.c_void_p * k)(*[x.ctypes.data for x in aa])
        pb = (ctypes.c_void_p * k)(*[x.ctypes.data for x in bb])
        seeds = None if filler_seeds is None else np.ascontiguousarray(np.asarray(filler_seeds, dtype=np.uint64))
        out = (ctypes.c_void_p * k)()
        st = lib.gl_prover_pool_prove_matmul(self.handle, k, pa, pb, _p(seeds) if seeds is not None else None, out)
        proofs = [Proof(out[i], self.host.n) if out[i] else None for i in range(k)]      # owned from here on: freed even when a lane failed
        check(st)
        return proofs

    def close(self):
        if self.handle:
            lib.gl_prover_pool_free(self.handle)
            self.handle = None


class CircuitView:
    """The same device-resident CircuitData used from another context (stream) of the same device."""

    def __init__(self, circuit_data, ctx):
        self.cd, self.ctx = circuit_data, ctx
        self.n = 1 << circuit_data.desc.degree_bits          # CircuitData and GenericCircuitData both carry the description

    def prove_device(self, d_wires_ptr, public_inputs, public_inputs_hash=None):
        return _prove_device(self.ctx, self.cd.handle, self.n, d_wires_ptr, public_inputs, public_inputs_hash)

    def prove(self, wires, public_inputs):
        wires, pis = _u64(wires), _u64(public_inputs)
        if wires.shape != (135, self.n):
            raise ValueError("wire matrix must be [135][n]")
        if pis.size == 0:
            pis = np.zeros(1, dtype=np.uint64)[:0]
        h = ctypes.c_void_p()
        check(lib.gl_prove(self.ctx.handle, self.cd.handle, _p(wires), _p(pis), pis.size, ctypes.byref(h)))
        return Proof(h.value, self.n)

    def prove_columns(self, columns, public_inputs):
        return _prove_columns(self.ctx, self.cd.handle, self.n, columns, public_inputs)


class FriProver:
    """fri_proof (fri/prover.rs:20-66) split at every transcript dependency."""

    def __init__(self, cd, batches, zeta, alpha, ctx):
        self.ctx, self.cd, self.batches = ctx, cd, list(batches)     # the batches must outlive the FRI state
        arr = (ctypes.c_void_p * 4)(*[b.handle for b in self.batches])
        h = ctypes.c_void_p()
        check(lib.gl_fri_combine(ctx.handle, cd.handle, arr, _p(_u64(zeta)), _p(_u64(alpha)), ctypes.byref(h)))
        self.handle = h.value

    def commit_round(self):
        cap = np.empty((1 << self.cd.desc.cap_height, 4), dtype=np.uint64)
        check(lib.gl_fri_commit_round(self.handle, _p(cap)))
        return cap

    def fold(self, beta):
        check(lib.gl_fri_fold(self.handle, _p(_u64(beta))))

    def final_poly(self):
        k = ctypes.c_size_t()
        check(lib.gl_fri_final_poly(self.handle, None, 0, ctypes.byref(k)))
        out = np.empty(k.value, dtype=np.uint64)
        check(lib.gl_fri_final_poly(self.handle, _p(out), out.size, ctypes.byref(k)))
        return out.reshape(-1, 2)

    def query(self, x_index):
        xi = np.ascontiguousarray(np.asarray(x_index, dtype=np.uint32))
        k = ctypes.c_size_t()
        check(lib.gl_fri_query(self.handle, _p(xi), xi.size, None, 0, ctypes.byref(k)))
        blob = np.empty(k.value, dtype=np.uint8)
        check(lib.gl_fri_query(self.handle, _p(xi), xi.size, _p(blob), blob.size, ctypes.byref(k)))
        return blob.tobytes()

    def __del__(self):
        try:
            if self.handle:
                lib.gl_fri_free(self.handle)
                self.handle = None
        except Exception:
            pass


class Challenger:
    """plonky2::iop::challenger::Challenger (iop/challenger.rs:30-153), host code."""

    def __init__(self):
        self.handle = lib.gl_challenger_new()

    def observe_elements(self, xs):
        a = _u64(np.asarray(xs, dtype=np.uint64).reshape(-1))
        check(lib.gl_challenger_observe(self.handle, _p(a), a.size))

    def get_n_challenges(self, n):
        out = np.empty(n, dtype=np.uint64)
        check(lib.gl_challenger_get_challenges(self.handle, _p(out), n))
        return [int(x) for x in out]

    def state(self):
        """(sponge_state[12], input_buffer) as fri_proof_of_work reads them."""
        st, buf, k = np.empty(12, dtype=np.uint64), np.empty(8, dtype=np.uint64), ctypes.c_uint32()
        check(lib.gl_challenger_state(self.handle, _p(st), _p(buf), ctypes.byref(k)))
        return st, buf[: k.value].copy()

    def __del__(self):
        try:
            if self.handle:
                lib.gl_challenger_free(self.handle)
                self.handle = None
        except Exception:
            pass


def pow_grind(sponge_state, input_buffer, min_leading_zeros, ctx=None):
    """fri_proof_of_work (fri/prover.rs:115-160): the smallest valid witness."""
    ctx = _ctx(ctx)
    st, buf = _u64(sponge_state), _u64(input_buffer)
    if st.size != 12:
        raise ValueError("sponge state has 12 words")
    w = np.zeros(1, dtype=np.uint64)
    check(lib.gl_pow_grind(ctx.handle, _p(st), _p(buf) if buf.size else None, buf.size, min_leading_zeros, _p(w)))
    return int(w[0])


class Proof:
    def __init__(self, handle, n):
        self.handle, self.n = handle, n

    def to_bytes(self):
        """ProofWithPublicInputs::to_bytes (plonk/proof.rs:104-110)."""
        k = lib.gl_proof_num_bytes(self.handle)
        buf = np.empty(k, dtype=np.uint8)
        check(lib.gl_proof_bytes(self.handle, _p(buf), k))
        return buf.tobytes()

    def challenges(self):
        out = np.zeros(64, dtype=np.uint64)
        k = lib.gl_proof_challenges(self.handle, _p(out))
        v = [int(x) for x in out[:k]]
        return {"betas": v[0:2], "gammas": v[2:4], "alphas": v[4:6], "zeta": v[6:8], "fri_alpha": v[8:10], "pow_witness": v[10],
                "public_inputs_hash": v[11:15], "fri_betas": [v[i:i + 2] for i in range(15, k, 2)]}

    def caps(self):
        out = np.empty((3, 16, 4), dtype=np.uint64)
        check(lib.gl_proof_caps(self.handle, _p(out)))
        return out

    def zs_partial_products(self, ncols=20):
        """Z and partial products as value columns; 34 columns for a circuit with lookups (the lookup polynomials follow)."""
        out = np.empty((ncols, self.n), dtype=np.uint64)
        check(lib.gl_proof_zs_partial_products(self.handle, _p(out)))
        return out

    def quotient_chunks(self):
        out = np.empty((16, self.n), dtype=np.uint64)
        check(lib.gl_proof_quotient_chunks(self.handle, _p(out)))
        return out

    def query_indices(self):
        out = np.zeros(256, dtype=np.uint64)
        k = lib.gl_proof_query_indices(self.handle, _p(out))
        return [int(x) for x in out[:k]]

    def __del__(self):
        try:
            if self.handle:
                lib.gl_proof_free(self.handle)
                self.handle = None
        except Exception:
            pass
