#!/usr/bin/env python3
"""Headline benchmark on MI355X: proofs/sec for the m=64 matmul circuit (full prove()), plus the 2^20 NTT leg.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU.  Proofs are independent: rank r proves its own
   random-witness proofs with replicated circuit data and no data-path collective; the only collective is one
   RCCL all_gather of the Merkle caps of the proved batch at the end of the timed region -> weak scaling.)

A "step" is one pass of the hot path over one unit of synthetic input resident in HBM: one prove()
(plonky2/src/plonk/prover.rs:102-329 from the full witness matrix on: 3 PolynomialBatch commitments, permutation
argument, quotient, openings, FRI) of the m = 64 circuit (n = 2^15 rows, 135 wire columns, 250 756-byte proof).
value = whole-job proofs per second = world * K / t.  Witness generation is excluded, the witness matrix is
already on the device when the clock starts.

The same JSON line carries
  roofline:     BASELINE configs[1], the HBM-bound kernel family of the path: forward 2^20-point NTT over 64 polynomials,
                algorithmic bytes 16*L*B (one 8-B read + one 8-B write per element) over the device time of
                its launches (HIP events on the launch stream);
  ntt:          GF elements/s of forward+inverse 2^20 NTTs (the second half of BASELINE.json's metric);
  cpu_baseline: the CPU restatement of the reference prover (oracle/, kind "port") proving the same circuit on the
                host cores (one proof, bounded).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
M = 64
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def synth_field(torch, shape, seed, device):
    """Canonical field elements as int64 bit patterns: hi word < 2^32 - 1 keeps the value below p."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hi = torch.randint(0, 2**32 - 1, shape, generator=g, device=device, dtype=torch.int64)
    lo = torch.randint(0, 2**32, shape, generator=g, device=device, dtype=torch.int64)
    return (hi << 32) | lo


def cpu_baseline(m):
    """BASELINE ONLY: times the oracle's prove() (C++ restatement of plonk/prover.rs) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    try:
        cores = len(os.sched_getaffinity(0))       # the CPU share this process may actually run on
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = min(cores, 64)
    oc = orc.circuit(m, threads=threads)
    a = oracle_lib.rand_field(64, m * m) % (2**32 - 1)
    b = oracle_lib.rand_field(65, m * m) % (2**32 - 1)
    w = oc.witness(a, b)
    t0 = time.perf_counter()
    proof = w.prove(threads=threads)
    dt = time.perf_counter() - t0
    ok = proof.verify()[0]
    return {
        "value": 1.0 / dt, "unit": "proofs/s", "cores": threads, "kind": "port",
        "sample": "1 proof of the m=%d circuit by the C++ restatement of the reference prover (oracle/gl_prover.hpp), %d threads over "
                  "columns / LDE points as the reference's Rayon axes, %.2f s wall, verifier restatement %s" % (m, threads, dt, "accepts" if ok else "REJECTS"),
    }


def ntt_leg(torch, ctx, lib, check, dev, batch):
    L = 1 << LOG_N
    data = synth_field(torch, (batch, L), 20, dev)
    ref = data.clone()
    ptr = ctypes.c_void_p(data.data_ptr())
    for _ in range(2):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, batch))
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, batch))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    intact = bool(torch.equal(data, ref))
    ctx.timing(True)
    for _ in range(reps):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch))
    rep = ctx.timing_report()
    ctx.timing(False)
    fwd_ms = sum(v["ms"] for v in rep.values()) / reps
    launches = {k: {"per_ntt": v["count"] / reps, "avg_ms": round(v["ms"] / v["count"], 5)} for k, v in rep.items()}
    algo = 16.0 * L * batch
    achieved = algo / (fwd_ms * 1e-3) / 1e9
    traffic = None
    tr_path = os.path.join(ROOT, "profiles", "ntt20_traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic = json.load(open(tr_path)).get("forward_ntt_hbm_bytes")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": "forward 2^20 NTT x %d polynomials = ntt_col_pass<10> + ntt_row_pass<10>" % batch,
                "algorithmic_bytes": algo, "launches": launches,
                "note": "VALU-issue-bound on gfx950 (no 64x64 multiplier): ~390 VALU instructions per element over the two passes at "
                        "the measured ~0.55 wave-instructions/ns/SIMD cap this instruction stream near 18 % of the HBM roof "
                        "(DESIGN.md section 4, profiles/README.md)"}
    ntt = {"metric": "Goldilocks NTT GF-elems/sec at 2^20 (forward+inverse)", "value": reps * 2.0 * batch * L / dt, "unit": "GF-elems/s",
           "batch": batch, "round_trip_bit_exact": intact}
    # the two reference points SURVEY 8(d) asks for next to the roofline fraction: a measured device copy (what "HBM-bound"
    # can reach on this box) and the integer-ALU rate of the kernel family that actually bounds prove()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    dst = torch.empty_like(data)
    dst.copy_(data)
    ev[0].record()
    for _ in range(5):
        dst.copy_(data)
    ev[1].record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2.0 * data.numel() * 8 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
    nperm = 1 << 23                                     # ~4 ms of work: host-clock timing around a context sync is accurate enough
    states = synth_field(torch, (nperm, 12), 21, dev)
    torch.cuda.synchronize()
    check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(states.data_ptr()), nperm))
    ctx.synchronize()
    t0 = time.perf_counter()
    check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(states.data_ptr()), nperm))
    ctx.synchronize()
    perm_per_s = nperm / (time.perf_counter() - t0)
    roofline["measured_copy_GBs"] = copy_gbs            # torch tensor copy of the same 512 MiB (read + write bytes)
    roofline["frac_of_measured_copy"] = achieved / copy_gbs
    ntt["poseidon"] = {"permutations_per_s": perm_per_s, "modular_multiplies_per_s": perm_per_s * 460,
                       "note": "one 12-word state per lane (k_poseidon_states, 2^23 states): the integer-ALU rate that bounds prove(); "
                               "18.9 k VALU instructions per permutation (profiles/README.md)"}
    del data, ref, dst, states
    return roofline, ntt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--m", type=int, default=M)
    ap.add_argument("--witnesses", type=int, default=4, help="distinct random witnesses cycled through")
    ap.add_argument("--ntt-batch", type=int, default=64)
    ap.add_argument("--streams", type=int, default=4, help="independent proofs in flight per GPU (one HIP stream + host thread each)")
    ap.add_argument("--e2e-steps", type=int, default=48, help="proofs of the secondary run that also times witness generation (0 = skip)")
    ap.add_argument("--e2e-lanes", type=int, default=0, help="proofs in flight of the secondary run (default: three times --streams)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run the RCCL path runs even for one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import plonky2_demo_amd as p
    from plonky2_demo_amd._lib import check, lib

    stream = torch.cuda.current_stream().cuda_stream
    ctx = p.Context(device=local_rank, stream=stream)
    m = args.m
    hc = p.MatmulCircuit(m)
    cd = hc.build(ctx)                                    # circuit data replicated on every GPU
    # independent proofs overlap on separate streams (the transcript forces ~10 host syncs inside one proof)
    import threading
    nstreams = max(1, args.streams)
    lanes = [(ctx, cd)] + [(lambda c: (c, p.api.CircuitView(cd, c)))(p.Context(device=local_rank)) for _ in range(nstreams - 1)]
    wit = []
    for k in range(args.witnesses):
        rng = np.random.default_rng(1000 * rank + k)        # u32 entries as matrix_mul.rs:76-78
        a = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)
        b = rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)
        wires, pis = hc.witness(a, b, filler_seed=1000 * rank + k)
        t = torch.from_numpy(wires.view(np.int64)).to(dev)  # witness matrix resident in HBM
        wit.append((t, pis))

    def step(i, lane=0):
        t, pis = wit[i % len(wit)]
        return lanes[lane][1].prove_device(ctypes.c_void_p(t.data_ptr()), pis)

    def run_steps(first, count, sink):
        """`count` proofs, round-robin over the lanes, one host thread per lane (ctypes releases the GIL)."""
        def work(lane):
            for i in range(first + lane, first + count, nstreams):
                sink[i - first] = step(i, lane).caps()
            lanes[lane][0].synchronize()
        ths = [threading.Thread(target=work, args=(k,)) for k in range(nstreams)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(0, max(args.warmup, nstreams), [None] * max(args.warmup, nstreams))
    if use_dist:                                            # untimed: brings the RCCL communicator up
        from plonky2_demo_amd import sharding
        sharding.gather_caps(np.zeros((1, 3, 16, 4), dtype=np.uint64), world, device=dev)
    barrier()
    caps = [None] * args.steps
    t0 = time.perf_counter()
    run_steps(0, args.steps, caps)
    gathered = None
    if use_dist:                                            # the Merkle-cap gather: 3 x 16 x 32 B per proof
        from plonky2_demo_amd import sharding
        # rank r proved global proofs r, r + world, ... (round-robin); every rank ends with all caps in proof order
        gathered = sharding.gather_caps(np.stack(caps), world * args.steps, device=dev)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        assert gathered.shape == (world * args.steps, 3, 16, 4)
        assert (gathered[rank::world] == np.stack(caps)).all()

    proof_bytes = len(step(0).to_bytes())
    roofline, ntt = (None, None)
    if rank == 0:
        # secondary figure (not `value`): the same loop with witness generation inside the clock -- operands on the host,
        # arithmetic rows filled by the GPU, the sequential public-input hash sponge by the lane's host thread
        e2e = None
        if args.e2e_steps > 0 and world == 1:                # single-GPU runs only: the scaling runs measure `value`
            # the host part of witness generation (several ms of sequential Poseidon per proof) is hidden by keeping three times
            # as many proofs in flight: while one lane's host thread hashes, the GPU works on the other lanes' proofs
            nl = args.e2e_lanes if args.e2e_lanes > 0 else 3 * nstreams
            el = lanes + [(lambda c: (c, p.api.CircuitView(cd, c)))(p.Context(device=local_rank)) for _ in range(nl - nstreams)]
            gens = [hc.witness_generator(c) for c, _ in el]
            bufs = [torch.empty((135, hc.n), dtype=torch.int64, device=dev) for _ in el]
            ops = [(np.random.default_rng(77 + k).integers(0, 2**32 - 1, m * m, dtype=np.uint64),
                    np.random.default_rng(177 + k).integers(0, 2**32 - 1, m * m, dtype=np.uint64)) for k in range(4)]

            def e2e_work(lane, count):
                for i in range(lane, count, nl):
                    a, b = ops[i % len(ops)]
                    ptr = ctypes.c_void_p(bufs[lane].data_ptr())
                    pis = gens[lane].run(a, b, ptr, filler_seed=i)
                    el[lane][1].prove_device(ptr, pis, gens[lane].public_inputs_hash)
                el[lane][0].synchronize()

            for count in (nl, args.e2e_steps):                # warm-up round, then the timed one
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ths = [threading.Thread(target=e2e_work, args=(k, count)) for k in range(nl)]
                [th.start() for th in ths]
                [th.join() for th in ths]
                torch.cuda.synchronize()
                t_e2e = time.perf_counter() - t1
            e2e = {"value": args.e2e_steps / t_e2e, "unit": "proofs/s", "proofs": args.e2e_steps, "ms_per_proof": t_e2e / args.e2e_steps * 1e3,
                   "proofs_in_flight": nl,
                   "includes": "witness generation from host operands (GPU arithmetic rows + host hash-sponge rows) + prove()"}
            del gens, bufs, el
        roofline, ntt = ntt_leg(torch, ctx, lib, check, dev, args.ntt_batch)
        ctx.timing(True)
        step(0)
        scopes = {k: round(v["ms"], 4) for k, v in ctx.timing_report().items()}
        ctx.timing(False)
        out = {
            "metric": "proofs/sec for m=64 matmul circuit (full prove(), PoseidonGoldilocksConfig, standard_recursion_config)",
            "value": world * args.steps / dt,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, 64-bit modular integer)",
            "data": "synthetic",
            "config": {"workload": "prove_matmul_m%d" % m, "trace_rows": hc.n, "lde_size": hc.n << 3, "proof_bytes": proof_bytes,
                       "witnesses_per_gpu": len(wit), "proofs_in_flight_per_gpu": nstreams, "parallelism": "independent proofs per GPU; RCCL all_gather of Merkle caps only"},
            "roofline": roofline,
            "ntt": ntt,
            "prove_device_ms_by_scope": scopes,
            "with_witness_generation": e2e,
        }
        out["cpu_baseline"] = cpu_baseline(m) if (world == 1 and not args.no_cpu) else None
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
