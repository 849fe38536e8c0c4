#!/usr/bin/env python3
"""Times gl_prove on the m x m matmul circuit and prints the per-scope device timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = p.default_context()
hc = p.MatmulCircuit(m)
rng = np.random.default_rng(1)
a = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64); b = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)
t = time.time(); wires, pis = hc.witness(a, b); tw = time.time() - t
t = time.time(); cd = hc.build(); ctx.synchronize(); tb = time.time() - t
d = ctx.alloc(wires.nbytes).upload(wires)
cd.prove_device(d.ptr, pis); ctx.synchronize()
t = time.time()
for _ in range(reps):
    pr = cd.prove_device(d.ptr, pis)
ctx.synchronize(); tp = (time.time() - t) / reps
print("m=%d n=2^%d witness %.3fs build(device) %.3fs prove %.2f ms (%.1f proofs/s) proof %d bytes" % (m, hc.degree_bits, tw, tb, tp * 1e3, 1 / tp, len(pr.to_bytes())))
# the same with public_inputs_hash supplied (as the witness generator does): no 3 m^2 / 8 sequential host permutations inside prove()
gen = hc.witness_generator(ctx)
pis2 = gen.run(a, b, d.ptr)
cd.prove_device(d.ptr, pis2, gen.public_inputs_hash); ctx.synchronize()
t = time.time()
for _ in range(reps):
    pr2 = cd.prove_device(d.ptr, pis2, gen.public_inputs_hash)
ctx.synchronize(); tp2 = (time.time() - t) / reps
print("     with the public-input hash supplied: prove %.2f ms (%.1f proofs/s)" % (tp2 * 1e3, 1 / tp2))
ctx.timing(True)
cd.prove_device(d.ptr, pis)
rep = ctx.timing_report()
ctx.timing(False)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-45s %4d launches %9.3f ms" % (k, v["count"], v["ms"]))
