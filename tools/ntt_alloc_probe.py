#!/usr/bin/env python3
"""Does the 2^20 x 64 NTT's time depend on WHERE its buffers lie?  The same launches with the data at different offsets inside one large
allocation (the inter-pass scratch stays where it is):  python tools/ntt_alloc_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
ctx = p.default_context()
lg, batch = 20, 64
nbytes = (batch << lg) * 8
def run(ptr, label):
    for _ in range(6): check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, lg, batch))
    ctx.synchronize(); ctx.timing(True)
    for _ in range(30): check(lib.gl_ntt_forward(ctx.handle, ptr, lg, batch))
    rep = ctx.timing_report(); ctx.timing(False)
    print("%-28s col %.4f  row %.4f  sum %.4f ms" % (label, *(rep[k]["ms"] / rep[k]["count"] for k in ("ntt_col_pass(forward)", "ntt_row_pass(forward)")), sum(v["ms"] for v in rep.values()) / 30), flush=True)
big = ctx.alloc(nbytes + (64 << 20))
x = np.random.default_rng(1).integers(0, 2**63, (batch, 1 << lg), dtype=np.uint64)
for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 32 << 20, (32 << 20) + 8192, 48 << 20):
    check(lib.gl_copy_h2d(ctx.handle, ctypes.c_void_p(big.ptr + off), x.ctypes.data_as(ctypes.c_void_p), nbytes))
    run(ctypes.c_void_p(big.ptr + off), "data at +%d" % off)
others = [ctx.alloc(s << 20) for s in (3, 17, 129, 300)]      # shift the next allocations
d2 = ctx.alloc(nbytes).upload(x)
run(ctypes.c_void_p(d2.ptr), "a later allocation")
