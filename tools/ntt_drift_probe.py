#!/usr/bin/env python3
"""The 2^20 x 64 forward NTT's time per block of 40 launches over a long run, with and without a burst of streaming kernels (a torch
randint + clone, what bench.py's leg starts with) in front:  python tools/ntt_drift_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
ctx = p.Context(device=0)
lg, batch = 20, 64
x = np.random.default_rng(1).integers(0, 2**63, (batch, 1 << lg), dtype=np.uint64)
d = ctx.alloc(x.nbytes).upload(x)
ptr = ctypes.c_void_p(d.ptr)
def block(label):
    ctx.timing(True)
    for _ in range(40): check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch))
    rep = ctx.timing_report(); ctx.timing(False)
    print("%-34s %.4f ms" % (label, sum(v["ms"] for v in rep.values()) / 40), flush=True)
t0 = time.time()
for k in range(6): block("block %d (t = %.2f s)" % (k, time.time() - t0))
a = torch.randint(0, 2**31, (64, 1 << 20), device="cuda:0", dtype=torch.int64); b = a.clone(); c = a.clone(); torch.cuda.synchronize()
for k in range(6): block("after torch streaming, block %d" % k)
time.sleep(3)
for k in range(3): block("after 3 s idle, block %d" % k)
