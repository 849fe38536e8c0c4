#!/usr/bin/env python3
"""Poseidon throughput on the device: python tools/time_poseidon.py
  * permutations/s of gl_poseidon_permute (one 12-word state per lane, 2^23 states), checked against the known-answer vectors
  * the leaf-hash launch of the m = 64 wires tree (2^18 leaves x 135 elements = 17 permutations per leaf), HIP events"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
ctx = p.default_context()
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
for kat in golden["poseidon_kats"]:
    got = p.poseidon(np.array(kat["input"], dtype=np.uint64))
    assert [int(x) for x in got] == kat["output"], "KAT mismatch"
n = 1 << 23
rng = np.random.default_rng(1)
st = rng.integers(0, 2**63, (n, 12), dtype=np.uint64)
d = ctx.alloc(st.nbytes).upload(st)
for _ in range(2):
    check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(d.ptr), n))
ctx.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(d.ptr), n))
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print("KATs ok; gl_poseidon_permute: %.3f ms for 2^23 states -> %.3f G permutations/s" % (dt * 1e3, n / dt / 1e9))
d.free()
# the wires tree of an m = 64 proof
cols = rng.integers(0, 2**63, (135, 1 << 15), dtype=np.uint64)
dc = ctx.alloc(cols.nbytes).upload(cols)
b = p.PolynomialBatch.from_device(dc.ptr, 135, 1 << 15, 3, 4, True, ctx=ctx)
ctx.timing(True)
for _ in range(3):
    b2 = p.PolynomialBatch.from_device(dc.ptr, 135, 1 << 15, 3, 4, True, ctx=ctx)
    b2.free()
rep = ctx.timing_report()
ctx.timing(False)
for k, v in rep.items():
    print("  %-28s %4d x  avg %.4f ms" % (k, v["count"], v["ms"] / v["count"]))
lh = rep["merkle_leaf_hash"]
print("leaf hash 2^18 x 135: %.3f ms -> %.3f G permutations/s" % (lh["ms"] / lh["count"], 17 * (1 << 18) / (lh["ms"] / lh["count"]) / 1e6))
