#!/usr/bin/env python3
"""Device time of the lookup-polynomial phase (compute_all_lookup_polys: k_lookup_inverses + the single-thread k_lookup_scan per table)
next to the whole proof, for circuits with one and two 256-entry tables:  python tools/lookup_time.py
(MI355X, round 3: 0.03-0.06 ms of a 2.3-2.6 ms proof at n = 2^7..2^9 -- 0.13-0.25 ms while the serial scan read all 78 wires of a row
itself; the per-row sums are now taken in parallel by k_lookup_inverses and the scan reads 8 words per row)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import plonky2_demo_amd as p, oracle_lib
orc = oracle_lib.load()
ctx = p.default_context()
for kind, param in ((8, 200), (8, 1000), (10, 400)):
    oc = orc.circuit_of_kind(kind, param, threads=8)
    n_in = param if kind == 8 else 2 * param
    w = oc.witness((np.arange(n_in, dtype=np.uint64) * 7) % 256, np.zeros(0, dtype=np.uint64), filler_seed=1)
    cd = p.GenericCircuitData(oc.product_desc(), oc.constants_sigmas())
    wires, pis = w.wires(), w.public_inputs()
    for _ in range(3): pr = cd.prove(wires, pis)
    ctx.timing(True)
    for _ in range(5): pr = cd.prove(wires, pis)
    rep = ctx.timing_report(); ctx.timing(False)
    tot = sum(v["ms"] for v in rep.values())
    print(kind, param, "n = 2^%d" % oc.info["degree_bits"], "lookup polys %.3f ms of %.3f ms per proof (sum of scopes)" % (rep["compute lookup polys"]["ms"] / 5, tot / 5))
