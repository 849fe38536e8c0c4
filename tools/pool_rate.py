#!/usr/bin/env python3
"""Proofs/s of the generic prover pool (gl_prover_pool_create_generic + gl_prover_pool_prove_columns: C++ lanes, host witnesses, the
pinned H2D rings) on the m = 64 matmul circuit:  python tools/pool_rate.py [count=480] [lanes=16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import plonky2_demo_amd as p
count = int(sys.argv[1]) if len(sys.argv) > 1 else 480
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = 64
hc = p.MatmulCircuit(m)
rng = np.random.default_rng(1)
wits = []
for k in range(4):
    a, b = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64), rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)
    wires, pis = hc.witness(a, b, filler_seed=k)
    wits.append(([wires[j].copy() for j in range(135)], pis))
pool = p.api.GenericProverPool(hc.desc, hc.constants_sigmas(), lanes=lanes)
ref = [pr.to_bytes() for pr in pool.prove_columns(wits)]           # warm-up batch (and the reference bytes)
batch = [wits[i % 4] for i in range(count)]
t0 = time.perf_counter()
proofs = pool.prove_columns(batch)
dt = time.perf_counter() - t0
ok = all(proofs[i].to_bytes() == ref[i % 4] for i in range(0, count, 7))
print("generic pool, %d lanes: %d proofs from host witnesses in %.3f s = %.1f proofs/s (sampled proofs identical: %s)" % (lanes, count, dt, count / dt, ok))
