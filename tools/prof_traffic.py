#!/usr/bin/env python3
"""Driver for the HBM-traffic PMC passes: a calibration kernel with known bytes in the same 8-byte-per-lane access
shape (gl_field_op canonicalise: reads 8n, writes 8n bytes) followed by forward 2^20 NTTs over B polynomials."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plonky2_demo_amd import Context
from plonky2_demo_amd._lib import check, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = Context(0, torch.cuda.current_stream().cuda_stream)
n = B << 20
hi = torch.randint(0, 2**32 - 1, (n,), device="cuda", dtype=torch.int64)
lo = torch.randint(0, 2**32, (n,), device="cuda", dtype=torch.int64)
d = (hi << 32) | lo
out = torch.empty_like(d)
p, po = ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(out.data_ptr())
for _ in range(reps):
    check(lib.gl_field_op(ctx.handle, 5, p, None, None, po, n))        # calibration: 8n read + 8n written
torch.cuda.synchronize()
for _ in range(reps):
    check(lib.gl_ntt_forward(ctx.handle, p, 20, B))
torch.cuda.synchronize()
print("done", n)
