#!/usr/bin/env python3
"""Small driver for rocprofv3: forward+inverse 2^20 NTTs over a batch (same kernels as bench.py): python3 tools/prof_ntt.py [batch] [reps].
With reps >= 20 the launches run back to back for long enough that rocprofv3 --stats averages the steady state (the first ~50 ms after a
gap run ~10 % slower, tools/ntt_drift_probe.py): 4 x reps untimed-equivalent launches precede nothing -- all launches are in the trace, so
the run is made long instead (reps x 10 pairs)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plonky2_demo_amd import Context
from plonky2_demo_amd._lib import check, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = Context(0, torch.cuda.current_stream().cuda_stream)
hi = torch.randint(0, 2**32 - 1, (B, 1 << 20), device="cuda", dtype=torch.int64)
lo = torch.randint(0, 2**32, (B, 1 << 20), device="cuda", dtype=torch.int64)
d = (hi << 32) | lo
p = ctypes.c_void_p(d.data_ptr())
for _ in range(reps * (10 if reps >= 20 else 1)):
    check(lib.gl_ntt_forward(ctx.handle, p, 20, B))
    check(lib.gl_ntt_inverse(ctx.handle, p, 20, B))
torch.cuda.synchronize()
print("done")
