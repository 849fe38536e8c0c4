#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU: a batch of independent proofs of the m=64 circuit from host operands, wall clock
including witness generation / witness upload and the final gather of caps and proof bytes on the host.

  python tools/batch512.py [proofs=512] [lanes=8]

Two variants are timed:
  device-witness : gl_matmul_witgen_run (GPU arithmetic rows + host sponge rows)  -> gl_prove_device
  host-witness   : gl_matmul_witness on the host, 35 MB pageable H2D inside gl_prove (the PCIe-inclusive figure)
"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import plonky2_demo_amd as p

nproofs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = 64
hc = p.MatmulCircuit(m)
ctx0 = p.Context(device=0)
cd = hc.build(ctx0)
lanes = [(ctx0, cd)] + [(lambda c: (c, p.api.CircuitView(cd, c)))(p.Context(device=0)) for _ in range(nl - 1)]


def operands(seed):          # seeds 0..511 as SURVEY 8d config 4
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64), rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)


ops = [operands(s) for s in range(nproofs)]


def run(kind):
    gens = [hc.witness_generator(c) for c, _ in lanes] if kind == "device" else None
    bufs = [c.alloc(135 * hc.n * 8) for c, _ in lanes] if kind == "device" else None
    caps, sizes = [None] * nproofs, [0] * nproofs

    def work(lane, first, count):
        for i in range(first + lane, first + count, nl):
            a, b = ops[i]
            if kind == "device":
                pis = gens[lane].run(a, b, bufs[lane].ptr, filler_seed=i)
                pr = lanes[lane][1].prove_device(bufs[lane].ptr, pis, gens[lane].public_inputs_hash)
            else:
                wires, pis = hc.witness(a, b, filler_seed=i)
                pr = lanes[lane][1].prove(wires, pis)
            caps[i] = pr.caps()
            sizes[i] = len(pr.to_bytes())                      # the proof bytes are gathered on the host
        lanes[lane][0].synchronize()

    def go(first, count):
        ths = [threading.Thread(target=work, args=(k, first, count)) for k in range(nl)]
        [t.start() for t in ths]
        [t.join() for t in ths]

    go(0, nl)                                                   # warm-up: pools, tables
    t0 = time.perf_counter()
    go(0, nproofs)
    dt = time.perf_counter() - t0
    assert all(s == 250756 for s in sizes)
    print("config 4, 1 GPU, %s witness: %d proofs in %.3f s = %.1f proofs/s (%.2f ms/proof), %d proofs in flight, %.1f MB of proofs gathered"
          % (kind, nproofs, dt, nproofs / dt, dt / nproofs * 1e3, nl, sum(sizes) / 1e6), flush=True)


run("device")
run("host")

# the same batch through the C++ prover pool (gl_prover_pool_prove_matmul): threads, streams and witness generation inside the library
pool = p.ProverPool(hc, lanes=nl)
pool.prove_matmul(ops[:nl], list(range(nl)))
t0 = time.perf_counter()
proofs = pool.prove_matmul(ops, list(range(nproofs)))
sizes = [len(pr.to_bytes()) for pr in proofs]
dt = time.perf_counter() - t0
print("config 4, 1 GPU, C++ prover pool: %d proofs in %.3f s = %.1f proofs/s (%.2f ms/proof), %d lanes, %.1f MB of proofs gathered"
      % (nproofs, dt, nproofs / dt, dt / nproofs * 1e3, nl, sum(sizes) / 1e6), flush=True)
pool.close()
