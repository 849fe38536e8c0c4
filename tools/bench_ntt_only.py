#!/usr/bin/env python3
"""bench.py's NTT leg on its own, several times in one process: the FIRST leg of a process reads ~10 % slower than the following ones
(0.645 against 0.568 ms per 2^20 x 64 transform).  `python tools/bench_ntt_only.py [preload]`: preload = ntt (1 s of library NTTs first),
torch (1 s of torch kernels first), none."""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
dev = torch.device("cuda:0")
pre = sys.argv[1] if len(sys.argv) > 1 else "none"
if pre == "ntt":
    c0 = p.Context(device=0)
    x = np.random.default_rng(1).integers(0, 2**63, (64, 1 << 20), dtype=np.uint64)
    d = c0.alloc(x.nbytes).upload(x)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20): check(lib.gl_ntt_forward(c0.handle, ctypes.c_void_p(d.ptr), 20, 64))
        c0.synchronize()
elif pre == "torch":
    a = torch.randint(0, 2**31, (64, 1 << 20), device=dev, dtype=torch.int64)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20): a = a * 3 + 1
        torch.cuda.synchronize()
for k in range(4):
    if pre == "empty_cache" and k == 2:
        torch.cuda.empty_cache()                      # leg 2 gets freshly allocated blocks again
    ctx = p.Context(device=0)
    roofline, ntt, extra, ceil = bench.ntt_leg(torch, ctx, lib, check, dev, 64, with_rows=False)
    print(pre, "leg", k, {kk: v["avg_ms"] for kk, v in roofline["launches"].items()}, round(roofline["frac"], 4), flush=True)
