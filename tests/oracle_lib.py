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
