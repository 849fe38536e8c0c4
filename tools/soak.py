#!/usr/bin/env python3
"""Soak: N proofs of the m=64 circuit on several streams at once; every proof must be byte-identical to the first proof of its
witness and is checked by the native verifier.   python tools/soak.py [proofs=1000] [lanes=4]"""
import ctypes, hashlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p

nproofs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = 64
hc = p.MatmulCircuit(m)
ctx0 = p.Context(device=0)
cd = hc.build(ctx0)
lanes = [(ctx0, cd)] + [(lambda c: (c, p.api.CircuitView(cd, c)))(p.Context(device=0)) for _ in range(nl - 1)]
wit = []
for k in range(6):
    rng = np.random.default_rng(k)
    a, b = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64), rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)
    wires, pis = hc.witness(a, b, filler_seed=k)
    buf = ctx0.alloc(wires.nbytes).upload(wires)
    ref = cd.prove_device(buf.ptr, pis).to_bytes()
    ok, why = cd.verify(ref)
    assert ok, why
    wit.append((buf, pis, hashlib.sha256(ref).digest()))
bad, done = [], [0] * nl


def work(lane):
    for i in range(lane, nproofs, nl):
        buf, pis, h = wit[i % len(wit)]
        by = lanes[lane][1].prove_device(buf.ptr, pis).to_bytes()
        if hashlib.sha256(by).digest() != h:
            bad.append((i, "bytes differ"))
        elif i % 7 == 0:
            ok, why = cd.verify(by)
            if not ok:
                bad.append((i, why))
        done[lane] += 1
    lanes[lane][0].synchronize()


t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(k,)) for k in range(nl)]
[t.start() for t in ths]
[t.join() for t in ths]
dt = time.perf_counter() - t0
print("soak: %d proofs on %d streams in %.1f s (%.1f proofs/s incl. hashing every proof and verifying every 7th): %d mismatches %s"
      % (sum(done), nl, dt, sum(done) / dt, len(bad), bad[:5]))
sys.exit(1 if bad else 0)
