#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json configs[1]): 2^20-point Goldilocks forward + inverse NTT on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; the NTT path shards by polynomial with no
   data-path collective -> weak scaling, each rank transforms its own batch.)

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM: an in-place forward
NTT of B polynomials of length 2^20 (fft_with_options, field/src/fft.rs:56-65) followed by the inverse
(ifft_with_options, :72-95).  value = GF elements transformed per second, whole job:
    world * K * 2 * B * 2^20 / t.
Prints ONE JSON line on rank 0, including
  roofline:     forward NTT, algorithmic bytes 16*L*B (one 8-B read + one 8-B write per element)
                over the measured device time of its two launches (HIP events on the launch stream);
  cpu_baseline: the CPU restatement of the reference algorithm (oracle/, kind "port") on a bounded sample.
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
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def synth_field(torch, shape, seed, device):
    """Canonical field elements as int64 bit patterns: hi word < 2^32 - 1 keeps the value below p."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hi = torch.randint(0, 2**32 - 1, shape, generator=g, device=device, dtype=torch.int64)
    lo = torch.randint(0, 2**32, shape, generator=g, device=device, dtype=torch.int64)
    return (hi << 32) | lo


def cpu_baseline(batch_cpu):
    """TEST/BASELINE ONLY: times the oracle's radix-2 FFT (restating field/src/fft.rs) on host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    cores = os.cpu_count() or 1
    x = oracle_lib.rand_field(5, (batch_cpu, 1 << LOG_N))
    p = x.ctypes.data_as(ctypes.c_void_p)
    n, b = ctypes.c_size_t(1 << LOG_N), ctypes.c_size_t(batch_cpu)
    orc.lib.orc_fft_mt(p, n, ctypes.c_size_t(min(batch_cpu, cores)), 0, ctypes.c_uint(cores))   # warm-up, partial
    x0 = x.copy()
    t0 = time.perf_counter()
    orc.lib.orc_fft_mt(p, n, b, 0, ctypes.c_uint(cores))
    orc.lib.orc_fft_mt(p, n, b, 1, ctypes.c_uint(cores))
    dt = time.perf_counter() - t0
    # only the first min(batch, cores) polynomials went through the warm-up transform first
    ok = bool((x[cores:] == x0[cores:]).all()) if batch_cpu > cores else True
    return {
        "value": 2.0 * batch_cpu * (1 << LOG_N) / dt,
        "unit": "GF-elems/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d polynomials of 2^20, forward+inverse, C++ restatement of field/src/fft.rs, %d threads over "
                  "polynomials (as oracle.rs:54 par_iter), %.2f s wall%s" % (batch_cpu, cores, dt, "" if ok else " (ROUND TRIP FAILED)"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="polynomials of 2^20 per GPU (64 -> 512 MiB, beyond the 256 MiB Infinity Cache)")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scratch-log", type=int, default=0, help="log2 of the inter-pass scratch in elements (0 = library default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from plonky2_demo_amd import Context
    from plonky2_demo_amd._lib import check, lib

    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(device=local_rank, stream=stream)
    if args.scratch_log:
        ctx.set_scratch_elems(1 << args.scratch_log)
    B, L = args.batch, 1 << LOG_N
    data = synth_field(torch, (B, L), 20 + rank, dev)
    ref = data.clone()
    ptr = ctypes.c_void_p(data.data_ptr())

    def step():
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, B))
        check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, B))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # forward followed by inverse is the identity: the timed work must have left the input bit-identical
    intact = bool(torch.equal(data, ref))

    # per-launch device time of the forward transform (HIP events on the launch stream)
    ctx.timing(True)
    reps = 5
    for _ in range(reps):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, B))
        check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, B))
    rep = ctx.timing_report()
    ctx.timing(False)
    kern = {k: v["ms"] / v["count"] for k, v in rep.items()}
    launches_per_ntt = {k: v["count"] / reps for k, v in rep.items()}
    fwd_ms = sum(kern[k] * launches_per_ntt[k] for k in kern if "forward" in k)
    algo_bytes = 16.0 * L * B
    achieved = algo_bytes / (fwd_ms * 1e-3) / 1e9 if fwd_ms > 0 else 0.0
    traffic = None
    tr_path = os.path.join(ROOT, "profiles", "ntt20_traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic = json.load(open(tr_path)).get("forward_ntt_hbm_bytes_per_launch_pair")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "Goldilocks NTT GF-elems/sec at 2^20 (forward+inverse)",
            "value": world * args.steps * 2.0 * B * L / dt,
            "unit": "GF-elems/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, 64-bit modular integer)",
            "data": "synthetic",
            "config": {"workload": "ntt_2^20_forward+inverse", "log_n": LOG_N, "batch_per_gpu": B,
                       "bytes_per_gpu": 8 * B * L, "round_trip_bit_exact": intact,
                       "parallelism": "independent polynomials per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "forward NTT = ntt_col_pass<10> + ntt_row_pass<10> (two launches)",
                         "algorithmic_bytes": algo_bytes,
                         "launch_ms": {k: round(v, 5) for k, v in kern.items()}},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not intact:
        raise SystemExit("round trip mismatch: forward+inverse NTT did not restore the input")


if __name__ == "__main__":
    main()
