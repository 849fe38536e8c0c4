#!/usr/bin/env python3
"""Headline benchmark on MI355X: proofs/sec for the m=64 matmul circuit (full prove()), plus the 2^20 NTT leg.

  python bench.py --gpus N --steps K --warmup W [--config4]

  N = 1:  runs in this process.
  N > 1:  one rank per GPU over RCCL.  Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the
          ranks run directly; launched plainly (`python bench.py --gpus N`), the parent starts the N ranks itself -- a fresh
          torch.distributed.run child, BEFORE anything in the parent touches the GPU -- waits, and relays rank 0's JSON line.
          Fewer than N visible devices is an error (exit 3), never a silent n_gpus = 1.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM: the 16 independent proofs a GPU keeps
in flight, each one prove() (plonky2/src/plonk/prover.rs:102-329 from the full witness matrix on: 3 PolynomialBatch commitments,
permutation argument, quotient, openings, FRI) of the m = 64 circuit (n = 2^15 rows, 135 wire columns, 250 756-byte proof).
Default mode (BASELINE configs[2], "scaling": "weak"): every rank proves K steps = 16 K proofs of its own random witnesses, 16 in
flight on 16 streams (the lanes run continuously, not batch by batch), witness matrices already in HBM when the clock starts;
value = world * 16 K / t proofs/s, ms_per_step = t / K.  (With a step of ONE proof the driver's `--steps 20` would time 20 proofs
on 16 lanes: a 70 ms region that never reaches the steady state.)  Proofs are independent: circuit data is
replicated, there is no data-path collective; the only collective is one RCCL all_gather of the Merkle caps of the proved batch
inside the timed region.

--config4 (BASELINE configs[3], "scaling": "strong"): a batch of 512 independent proofs (operand seeds 0..511), proof i on rank
i mod N, through the library's prover pool: operands on the host, witness generation in HBM and prove() per lane inside the
clock, then the cap gather; value = 512 / t.  At N = 1 the default run reports it as the extra key "config4_batch512".

The same JSON line carries
  roofline:        BASELINE configs[1], the HBM-bound kernel family of the path: forward 2^20-point NTT over 64 polynomials,
                   algorithmic bytes 16*L*B (one 8-B read + one 8-B write per element) over the device time of
                   its launches (HIP events on the launch stream); valu_frac = the same launches against the VALU issue roof
                   (instructions per element from the committed SQ_INSTS_VALU pass), at the nominal and at the measured clock;
  roofline_prove:  the headline kernel family: Poseidon permutations per second of the proofs against the measured
                   one-state-per-lane ceiling of the same chip, and the 2.0 GB of algorithmic bytes per proof against HBM;
  ntt:             GF elements/s of forward+inverse 2^20 NTTs (the second half of BASELINE.json's metric);
  extra:           BASELINE.md section 3's other rows (NTT at batch 1 / 16 / 256, LDE 2^17 -> 2^20 x 135, Merkle commit 2^18 x 135);
  with_host_witness: the same 16 lanes through gl_prove_columns from HOST witness matrices (135 pageable vectors, 35 MB of H2D per
                   proof inside the clock): the PCIe-inclusive rate of the drop-in entry, never `value`;
                   `--config4 --host-witness` is BASELINE configs[3] in that form ("incl. H2D of witnesses");
  host:            lanes per GPU (capped by min(affinity mask, cgroup CPU quota) / ranks), GPU_MAX_HW_QUEUES, host CPU seconds;
  cpu_baseline:    the CPU restatement of the reference prover (oracle/, kind "port") on the host cores: all cores (median of 5
                   proofs after a warm-up); `one_thread` quotes the committed measurement of one full 1-thread proof (45 s);
  vs_baseline:     value / cpu_baseline.value -- the ratio to that CPU PORT (BASELINE.md publishes no number for this metric).
The last proof of every lane of the timed loop is verified after the clock stops (config.lanes_verified_after_the_clock).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The prover keeps 4 (12 in the batch
# mode) independent proofs in flight on as many streams; once RCCL adds its own streams two proofs share a queue and serialise
# (measured: 246 -> 212 proofs/s with a one-rank process group, 246 again with 8 queues; profiles/README.md).  Must be set
# before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # 16 proofs in flight + RCCL's own streams (16 queues: 270 proofs/s under RCCL, 20-24: 287)

LOG_N = 20
M = 64
VALU_CLOCK_HZ = 2.4e9          # MI355X peak engine clock (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
ALGO_BYTES_PER_PROOF_M64 = 2.0e9


def synth_field(torch, shape, seed, device):
    """Canonical field elements as int64 bit patterns: hi word < 2^32 - 1 keeps the value below p."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hi = torch.randint(0, 2**32 - 1, shape, generator=g, device=device, dtype=torch.int64)
    lo = torch.randint(0, 2**32, shape, generator=g, device=device, dtype=torch.int64)
    return (hi << 32) | lo


def permutations_per_proof(n, rate_bits=3, cap_height=4, arity_bits=4, pow_bits=16):
    """Poseidon permutations of one prove() of this circuit family: leaves N * ceil(C / 8) + N - 2^cap
    nodes for the wires (135), Z/partial products (20) and quotient (16) trees, the FRI round trees (leaves of 32 elements),
    one PoseidonGate evaluation per LDE point in the quotient, and the expected 2^pow_bits candidates of the grind."""
    N = n << rate_bits
    total = 0
    for cols in (135, 20, 16):
        total += N * ((cols + 7) // 8) + N - (1 << cap_height)
    lg = N.bit_length() - 1
    k = n
    while k > (1 << arity_bits) * 4 and lg - arity_bits >= cap_height:       # reduction_strategies.rs:39-49 (ConstantArityBits(4, 5))
        lg -= arity_bits
        leaves = 1 << lg
        total += leaves * 4 + leaves - (1 << cap_height)
        k >>= arity_bits
    return total + N + (1 << pow_bits)


def cpu_quota_cores():
    """CPU time the cgroup grants per period, in cores (cgroup v2 cpu.max), or None: the affinity mask of a container often shows
    every core of the host while the quota is a fraction of them."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def host_cores():
    try:
        return len(os.sched_getaffinity(0))       # the CPU share this process may actually run on
    except AttributeError:
        return os.cpu_count() or 1


def usable_cores():
    """Cores this process can really burn: the smaller of the affinity mask and the cgroup's CPU quota (a 1-GPU box shows 256
    cores in the mask and grants 16)."""
    q = cpu_quota_cores()
    n = float(host_cores())
    return min(n, q) if q else n


def lanes_per_rank(world, wanted):
    """Every proof in flight has a host thread that drives its ~10 transcript round trips and polls its stream (~0.3 of a core
    each once the waits sleep): with N ranks on one host the lanes of a rank are capped by the rank's share of the usable cores
    (never below 4).  Returns (cores per rank, {name: (wanted, granted)})."""
    per_rank = usable_cores() / max(1, world)
    cap = max(4, int(per_rank))
    return per_rank, {k: (v, min(v, cap)) for k, v in wanted.items()}


def cpu_micro_rows(orc, oracle_lib, threads):
    """BASELINE.md section 3's CPU columns for the micro-kernel rows, on bounded samples (opt-in: --cpu-rows)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    L = 1 << 20

    def med3(fn):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]
    x1 = oracle_lib.rand_field(70, (1, L))
    t_fft1 = med3(lambda: orc.fft(x1))
    xb = [oracle_lib.rand_field(71 + i, (1, L)) for i in range(threads)]
    with ThreadPoolExecutor(threads) as ex:                       # ctypes releases the GIL: one transform per core
        t_fftn = med3(lambda: list(ex.map(orc.fft, xb)))
    c1 = oracle_lib.rand_field(80, (2, 1 << 17))
    t_lde1 = med3(lambda: orc.lde(c1, 3, threads=1)) / 2 * 135
    cn = oracle_lib.rand_field(81, (max(2, min(threads, 135)), 1 << 17))
    t_lden = med3(lambda: orc.lde(cn, 3, threads=threads)) / cn.shape[0] * 135
    leaves = oracle_lib.rand_field(82, (1 << 14, 135))                                   # 1/16 of the 2^18 leaves
    t0 = time.perf_counter(); orc.merkle(leaves, 4); t_mk1 = (time.perf_counter() - t0) * 16
    perms = (1 << 18) * 17 + (1 << 18) - 16
    return {
        "ntt_2^20": {"one_thread_GF_elems_per_s": L / t_fft1, "all_cores_GF_elems_per_s": threads * L / t_fftn,
                     "one_thread_GBs": 16.0 * L / t_fft1 / 1e9, "all_cores_GBs": 16.0 * L * threads / t_fftn / 1e9,
                     "sample": "1 / %d polynomials of 2^20, median of 3" % threads},
        "coset_lde_2^17_to_2^20_x135": {"one_thread_s": t_lde1, "all_cores_s": t_lden, "one_thread_GBs": 72.0 * (1 << 17) * 135 / t_lde1 / 1e9,
                                        "all_cores_GBs": 72.0 * (1 << 17) * 135 / t_lden / 1e9,
                                        "sample": "2 columns (1 thread) / %d columns (%d threads), scaled to 135" % (cn.shape[0], threads)},
        "merkle_commit_2^18_x135_cap4": {"one_thread_s": t_mk1, "one_thread_permutations_per_s": perms / t_mk1,
                                         "sample": "2^14 of the 2^18 leaves on 1 thread, scaled x16"},
        "threads": threads,
    }


def cpu_baseline(m, rows=False, one_thread_full=False):
    """BASELINE ONLY: times the oracle's prove() (C++ restatement of plonk/prover.rs) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    cores = host_cores()
    threads = max(1, min(cores, 64))
    oc = orc.circuit(m, threads=threads)
    a = oracle_lib.rand_field(64, m * m) % (2**32 - 1)
    b = oracle_lib.rand_field(65, m * m) % (2**32 - 1)
    w = oc.witness(a, b)
    times = []
    ok = True
    for k in range(6):                              # 1 warm-up + 5 timed repetitions, median
        t0 = time.perf_counter()
        proof = w.prove(threads=threads)
        dt = time.perf_counter() - t0
        if k > 0 or dt > 12:
            times.append(dt)
        if dt > 12:                                 # slow host: one run is the bounded sample
            break
    ok = proof.verify()[0]
    med = sorted(times)[len(times) // 2]
    # 1 thread (the reference binary's configuration, matrix_mul.rs:19): a full 1-thread proof takes minutes, so the parallel
    # speed-up is measured on a bounded sample of the same proof -- the Z / partial-products commitment (PolynomialBatch::
    # from_values of 20 columns x n, rate 3, cap 4: iFFT + LDE + Poseidon Merkle tree) -- with 1 thread and with all threads
    micro = cpu_micro_rows(orc, oracle_lib, threads) if rows else None
    measured_1t = None
    if one_thread_full:                             # opt-in (--cpu-one-thread): one full proof on ONE thread, about a minute
        t0 = time.perf_counter(); p1 = w.prove(threads=1); t_full1 = time.perf_counter() - t0
        measured_1t = {"value": 1.0 / t_full1, "unit": "proofs/s", "seconds": round(t_full1, 2), "verifies": bool(p1.verify()[0])}
    return {
        "value": 1.0 / med, "unit": "proofs/s", "cores": threads, "threads": threads, "cores_affinity": cores, "cpu_quota_cores": cpu_quota_cores(), "kind": "port", "micro_kernel_rows": micro, "one_thread_measured": measured_1t,
        "runs_s": [round(t, 3) for t in times],
        "one_thread": committed_one_thread(m),
        "sample": "all cores: median of %d full proofs (after one warm-up proof) of the m=%d circuit by the C++ restatement of the reference prover (oracle/gl_prover.hpp), "
                  "%d threads on %d cores of the affinity mask (cgroup quota: %s cores) over the reference's Rayon axes, verifier restatement %s; "
                  "1 thread: not timed in the default run (a full 1-thread proof takes 45 s: --cpu-one-thread), `one_thread` quotes the committed measurement"
                  % (len(times), m, threads, cores, cpu_quota_cores(), "accepts" if ok else "REJECTS"),
    }


def committed_one_thread(m):
    """The measured 1-thread figure of the CPU port (one full proof, --cpu-one-thread), as committed under profiles/: quoted, not
    derived (round 2's derived figure was 1.8x off)."""
    for name in ("r03_cpu_one_thread.json", "r02_cpu_one_thread.json"):
        path = os.path.join(ROOT, "profiles", name)
        if m == 64 and os.path.exists(path):
            try:
                d = json.load(open(path))
                d = d.get("cpu_baseline", d)
                d = d.get("one_thread_measured") or d
                return {"value": d["value"], "unit": "proofs/s", "seconds": d.get("seconds"), "measured": True,
                        "source": "profiles/%s (committed profile of `bench.py --cpu-one-thread`, not measured in this run)" % name}
            except Exception:
                pass
    return None


def timed_launches(ctx, fn, reps, run_in=0):
    """`reps` timed calls of fn (HIP events per launch on the context's stream), directly behind `run_in` untimed ones: after ANY gap in
    the work (a host synchronisation, a torch kernel, an idle moment) the chip needs ~50 ms of the same launches to return to its
    clocks -- the first 40 launches of the 2^20 x 64 NTT read 0.632 ms, the following ones 0.565 (tools/ntt_drift_probe.py)."""
    for _ in range(run_in):
        fn()
    ctx.timing(True)
    for _ in range(reps):
        fn()
    rep = ctx.timing_report()
    ctx.timing(False)
    return rep


def ntt_leg(torch, ctx, lib, check, dev, batch, with_rows=True):
    L = 1 << LOG_N
    data = synth_field(torch, (batch, L), 20, dev)
    ref = data.clone()
    torch.cuda.synchronize()                 # the library runs on its own stream: the inputs must be complete first
    ptr = ctypes.c_void_p(data.data_ptr())
    # steady state: a cold GPU needs tens of milliseconds of work to reach its clocks (5 repetitions after 2 warm-up pairs
    # measured 0.85 ms per transform where 40 measure 0.75 ms; the prover itself keeps the GPU busy all the time)
    for _ in range(10):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, batch))
    ctx.synchronize()
    reps = 40
    t0 = time.perf_counter()
    for _ in range(reps):
        check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch)); check(lib.gl_ntt_inverse(ctx.handle, ptr, LOG_N, batch))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    intact = bool(torch.equal(data, ref))
    rep = timed_launches(ctx, lambda: check(lib.gl_ntt_forward(ctx.handle, ptr, LOG_N, batch)), reps, run_in=4 * reps)
    fwd_ms = sum(v["ms"] for v in rep.values()) / reps
    launches = {k: {"per_ntt": v["count"] / reps, "avg_ms": round(v["ms"] / v["count"], 5)} for k, v in rep.items()}
    algo = 16.0 * L * batch
    achieved = algo / (fwd_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tr_path = os.path.join(ROOT, "profiles", "ntt20_traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic = json.load(open(tr_path)).get("forward_ntt_hbm_bytes")
            traffic_src = "profiles/ntt20_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_traffic.py (committed profile, not measured in this run)"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "forward 2^20 NTT x %d polynomials = ntt_col_pass<10> + ntt_row_pass<10>" % batch,
                "algorithmic_bytes": algo, "launches": launches}
    # the kernel's own roof is VALU issue (gfx950: no 64x64 multiplier, no 64-bit add with carry-out): instructions per element from the
    # committed SQ_INSTS_VALU pass, one wave64 instruction per 4 cycles per SIMD at the nominal clock
    vi_path = os.path.join(ROOT, "profiles", "ntt20_valu.json")
    if os.path.exists(vi_path) and batch == 64:
        try:
            vi = json.load(open(vi_path))
            per_elem = vi["valu_instructions_per_element"]["col"] + vi["valu_instructions_per_element"]["row"]
            peak = 256 * 4 * VALU_CLOCK_HZ / 4.0
            roofline["valu_frac"] = per_elem * batch * L / 64.0 / peak / (fwd_ms * 1e-3)
            if "clock_GHz" in vi:
                # at the clock the SQ counters show for these kernels (SQ_BUSY_CYCLES / duration, ~2.0 GHz under load, not the nominal 2.4)
                clk = 0.5 * (vi["clock_GHz"]["col"] + vi["clock_GHz"]["row"]) * 1e9
                roofline["valu_frac_at_measured_clock"] = per_elem * batch * L / 64.0 / (256 * 4 * clk / 4.0) / (fwd_ms * 1e-3)
                roofline["measured_clock_GHz"] = clk / 1e9
                roofline["valu_utilisation_from_counters"] = vi["valu_utilisation_at_that_clock"]
            roofline["valu"] = {"instructions_per_element": per_elem, "peak_wave_instructions_per_s": peak,
                                "source": "profiles/ntt20_valu.json: SQ_INSTS_VALU of tools/prof_ntt.py (committed profile, not measured in this run); "
                                          "peak = 1024 SIMDs x %.1f GHz / 4 cycles per wave64 instruction" % (VALU_CLOCK_HZ / 1e9)}
            roofline["note"] = ("VALU-issue-bound: %.0f VALU instructions per element over the two passes (SQ_INSTS_VALU); "
                                "the memory-only / VALU-only times of the passes and the variants tried are in profiles/README.md" % per_elem)
        except Exception:
            pass
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
    nperm = 1 << 23                                     # ~3 ms of work: host-clock timing around a context sync is accurate enough
    states = synth_field(torch, (nperm, 12), 21, dev)
    torch.cuda.synchronize()
    check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(states.data_ptr()), nperm))
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        check(lib.gl_poseidon_permute(ctx.handle, ctypes.c_void_p(states.data_ptr()), nperm))
    ctx.synchronize()
    perm_per_s = 3 * nperm / (time.perf_counter() - t0)
    roofline["measured_copy_GBs"] = copy_gbs            # torch tensor copy of the same 512 MiB (read + write bytes)
    roofline["frac_of_measured_copy"] = achieved / copy_gbs
    ntt["poseidon"] = {"permutations_per_s": perm_per_s, "modular_multiplies_per_s": perm_per_s * 472,
                       "note": "one 12-word state per lane (k_poseidon_states, 2^23 states): the integer-ALU rate that bounds prove()"}
    del dst, states, ref
    if not with_rows:
        del data
        return roofline, ntt, None, perm_per_s
    # the other micro-kernel rows
    extra = {}
    for b2 in (1, 16, 256):
        d2 = data[:b2] if b2 <= batch else synth_field(torch, (b2, L), 22, dev)
        torch.cuda.synchronize()
        p2 = ctypes.c_void_p(d2.data_ptr())
        check(lib.gl_ntt_forward(ctx.handle, p2, LOG_N, b2)); check(lib.gl_ntt_inverse(ctx.handle, p2, LOG_N, b2))
        n2 = 40 if b2 <= 16 else 10
        r2 = timed_launches(ctx, lambda: (check(lib.gl_ntt_forward(ctx.handle, p2, LOG_N, b2)), check(lib.gl_ntt_inverse(ctx.handle, p2, LOG_N, b2))), n2, run_in=n2)
        f_ms = sum(v["ms"] for k, v in r2.items() if "forward" in k) / n2
        i_ms = sum(v["ms"] for k, v in r2.items() if "inverse" in k) / n2
        extra["ntt_2^20_batch_%d" % b2] = {"forward_ms": round(f_ms, 4), "inverse_ms": round(i_ms, 4), "forward_GBs": 16.0 * L * b2 / f_ms / 1e6,
                                           "GF_elems_per_s_fwd_inv": 2.0 * L * b2 / ((f_ms + i_ms) * 1e-3)}
        del d2
    del data
    lg, cols = 17, 135                                                      # benches/ffts.rs:21-37 shape at the config-2 size
    co = synth_field(torch, (cols, 1 << lg), 23, dev)
    out = torch.empty((cols, 1 << (lg + 3)), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    call = lambda: check(lib.gl_ntt_coset_lde(ctx.handle, ctypes.c_void_p(co.data_ptr()), lg, 3, cols, ctypes.c_void_p(out.data_ptr())))
    call()
    r3 = timed_launches(ctx, call, 10, run_in=20)
    ms = sum(v["ms"] for v in r3.values()) / 10
    extra["coset_lde_2^17_to_2^20_x135"] = {"ms": round(ms, 4), "algorithmic_GBs": 72.0 * (1 << lg) * cols / ms / 1e6, "frac_of_hbm_peak": 72.0 * (1 << lg) * cols / ms / 1e6 / HBM_PEAK_GBS}
    del co, out
    import plonky2_demo_amd as p
    vals = synth_field(torch, (135, 1 << 15), 24, dev)                      # benches/merkle.rs:12-26 shape at the m = 64 proof's size
    torch.cuda.synchronize()
    mk = lambda: p.PolynomialBatch.from_device(vals.data_ptr(), 135, 1 << 15, 3, 4, True, ctx=ctx).free()
    mk()
    r4 = timed_launches(ctx, mk, 10, run_in=10)
    leaf = r4["merkle_leaf_hash"]["ms"] / 10
    lev = r4["merkle_levels"]["ms"] / 10
    N = 1 << 18
    perms = N * 17 + N - 16
    extra["merkle_commit_2^18_x135_cap4"] = {"leaf_hash_ms": round(leaf, 4), "levels_ms": round(lev, 4), "permutations": perms,
                                             "permutations_per_s": perms / ((leaf + lev) * 1e-3), "leaf_read_GBs": 8.0 * 135 * N / leaf / 1e6}
    del vals
    return roofline, ntt, extra, perm_per_s


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Parent of a plain `python bench.py --gpus N`: starts the N ranks in a fresh torch.distributed.run child.  Nothing here
    initialises the GPU (device_count() only counts), so the children are not forked from / exec'ed over a GPU process."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (n, have))
        return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line)
    elif rc == 0:
        rc = 4
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cpu-one-thread", action="store_true", help="cpu_baseline also times one full proof on one thread (about a minute)")
    ap.add_argument("--cpu-rows", action="store_true", help="cpu_baseline also times the CPU columns of BASELINE.md section 3's micro-kernel rows (adds ~20 s)")
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step is one batch of --streams (16) proofs per GPU")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps (at least one)")
    ap.add_argument("--m", type=int, default=M)
    ap.add_argument("--witnesses", type=int, default=4, help="distinct random witnesses cycled through")
    ap.add_argument("--ntt-batch", type=int, default=64)
    ap.add_argument("--streams", type=int, default=16, help="independent proofs in flight per GPU (one HIP stream + host thread each)")
    ap.add_argument("--e2e-steps", type=int, default=160, help="proofs of the secondary run that also times witness generation (0 = skip)")
    ap.add_argument("--e2e-lanes", type=int, default=0, help="proofs in flight of the secondary run (default: --streams)")
    ap.add_argument("--config4", action="store_true", help="BASELINE configs[3]: one batch of --batch proofs sharded over the ranks (strong scaling)")
    ap.add_argument("--host-witness", action="store_true", help="with --config4: the witnesses are full host matrices (135 vectors of n, as the reference holds them) "
                    "and go through gl_prove_columns: the 35 MB H2D per proof is inside the clock (SURVEY 8d: 'incl. H2D of witnesses')")
    ap.add_argument("--host-witness-steps", type=int, default=160, help="proofs of the secondary run through gl_prove_columns (host witness, H2D in the clock; 0 = skip)")
    ap.add_argument("--batch", type=int, default=512, help="proofs of the config-4 batch")
    ap.add_argument("--pool-lanes", type=int, default=16, help="proofs in flight per GPU in the config-4 batch")
    ap.add_argument("--spawn", action="store_true", help="run even a 1-GPU job as a spawned rank over RCCL (the N > 1 code path)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other micro-kernel rows (NTT at batch 1/16/256, LDE, Merkle commit) and the config-4 batch")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ          # under torch.distributed.run the RCCL path runs even for one rank
    if os.environ.get("BENCH_SKIP_RCCL") and world == 1:
        use_dist = False                     # diagnosis only: a spawned rank without the process group (profiles/README.md)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        affinity = os.sched_getaffinity(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        after = os.sched_getaffinity(0)
        if after != affinity:
            # RCCL pins the initialising thread to the cores next to its GPU and the prover's host threads would inherit the
            # narrowed mask: give the process its CPU share back
            sys.stderr.write("bench.py: rank %d: CPU affinity %d -> %d cores after the RCCL init, restored\n" % (rank, len(affinity), len(after)))
            os.sched_setaffinity(0, affinity)

    # every proof in flight has a host thread that drives its ~10 transcript round trips (and polls its stream): with N ranks on
    # one host the lanes per rank are capped by the rank's share of the usable cores (never below 4)
    cores_per_rank, lane_caps = lanes_per_rank(world, {"streams": args.streams, "pool_lanes": args.pool_lanes})
    for name, (want, got) in lane_caps.items():
        if got != want:
            setattr(args, name, got)
            if rank == 0:
                sys.stderr.write("bench.py: --%s %d -> %d (%.1f usable cores per rank: min(affinity mask, cgroup quota) / ranks)\n" % (name.replace("_", "-"), want, got, cores_per_rank))

    import plonky2_demo_amd as p
    from plonky2_demo_amd import sharding
    from plonky2_demo_amd._lib import check, lib

    # Every context, lane 0's too, runs on a stream the library creates (non-blocking).  torch's current stream is a BLOCKING stream: a
    # launch on it is ordered against the null stream, which cost the 2^20 NTT 10 % (0.639 against 0.575 ms per transform for the very
    # same launches, tools/bench_ntt_only.py).  torch only synthesises inputs here, and every hand-over is behind a torch.cuda.synchronize().
    ctx = p.Context(device=local_rank)
    m = args.m
    hc = p.MatmulCircuit(m)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def operands(i):
        rng = np.random.default_rng(i)              # u32 entries as matrix_mul.rs:76-78
        return rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64), rng.integers(0, 2**32 - 1, m * m, dtype=np.uint64)

    def run_config4(count, lanes):
        """`count` proofs, proof i on rank i mod world, through gl_prover_pool_prove_matmul; returns (seconds, proof bytes checked)."""
        mine = sharding.proofs_for_rank(count, rank, world)
        ops = [operands(i) for i in mine]
        pool = p.ProverPool(hc, lanes=lanes, device=local_rank)
        try:
            warm = pool.prove_matmul(ops[: max(1, min(len(ops), lanes))], mine[: max(1, min(len(ops), lanes))])      # streams, allocators, tables
            del warm
            if use_dist:
                sharding.gather_caps(np.zeros((len(mine), 3, 16, 4), dtype=np.uint64), count, device=dev)             # brings the communicator up
            barrier()
            t0 = time.perf_counter()
            proofs = pool.prove_matmul(ops, mine)
            caps = np.stack([pr.caps() for pr in proofs]) if proofs else np.zeros((0, 3, 16, 4), dtype=np.uint64)
            allcaps = sharding.gather_caps(caps, count, device=dev) if use_dist else caps
            barrier()
            dt = time.perf_counter() - t0
            assert allcaps.shape == (count, 3, 16, 4) and (allcaps[rank::world] == caps).all()
            nbytes = len(proofs[0].to_bytes()) if proofs else 0
            ok = hc.verify(proofs[0].to_bytes(), pool.constants_sigmas_cap, pool.circuit_digest)[0] if proofs else True
        finally:
            pool.close()
        if use_dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, nbytes, ok

    def host_witness_run(cd_or_views, count, nl, nwit, seed0):
        """`count` proofs through gl_prove_columns on `nl` lanes: every proof's 135 host vectors (35 MB at m = 64) cross PCIe inside the
        clock.  Returns (seconds, host CPU seconds of the whole process, proofs, HostColumns used)."""
        hw = []
        for k in range(nwit):
            a, b = operands(seed0 + k)
            wires, pis = hc.witness(a, b, filler_seed=seed0 + k)
            hw.append((p.api.HostColumns([wires[c] for c in range(135)], hc.n), pis))
        res = [None] * count

        def work(lane, n_):
            for i in range(lane, n_, nl):
                cols, pis = hw[i % nwit]
                res[i] = cd_or_views[lane][1].prove_columns(cols, pis)
            cd_or_views[lane][0].synchronize()

        for n_ in (nl, count):                               # warm-up round (pinned chunks, pools), then the timed one
            barrier()
            c0 = time.process_time()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(k, n_)) for k in range(nl)]
            [th.start() for th in ths]
            [th.join() for th in ths]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cpu = time.process_time() - c0
        return dt, cpu, res, hw

    common = {"unit": "proofs/s", "n_gpus": world, "higher_is_better": True, "vs_baseline": None,
              "dtype": "u64 (Goldilocks, 64-bit modular integer)", "data": "synthetic",
              "host": {"lanes_per_gpu": args.streams, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "cores_affinity": host_cores(), "cpu_quota_cores": cpu_quota_cores(), "usable_cores_per_rank": usable_cores() / max(1, world)}}
    import threading

    if args.config4 and args.host_witness:
        # BASELINE configs[3] as SURVEY 8d words it: wall clock incl. the H2D of the witnesses.  Proof i on rank i mod N, each rank
        # drives its lanes through gl_prove_columns from full host witness matrices (cycled from 16 distinct ones), then the cap gather.
        mine = sharding.proofs_for_rank(args.batch, rank, world)
        cd4 = hc.build(ctx)
        nl = max(1, args.pool_lanes)
        views = [(ctx, cd4)] + [(lambda c: (c, p.api.CircuitView(cd4, c)))(p.Context(device=local_rank)) for _ in range(nl - 1)]
        if use_dist:
            sharding.gather_caps(np.zeros((len(mine), 3, 16, 4), dtype=np.uint64), args.batch, device=dev)
        cpu0 = time.process_time()
        dt, cpu_s, res, hw = host_witness_run(views, len(mine), nl, 16, 5000 * rank)
        t1 = time.perf_counter()
        caps = np.stack([pr.caps() for pr in res]) if res else np.zeros((0, 3, 16, 4), dtype=np.uint64)
        allcaps = sharding.gather_caps(caps, args.batch, device=dev) if use_dist else caps
        barrier()
        dt += time.perf_counter() - t1
        assert allcaps.shape == (args.batch, 3, 16, 4) and (allcaps[rank::world] == caps).all()
        ok = all(cd4.verify(res[i])[0] for i in range(0, len(res), max(1, len(res) // nl)))       # one proof per lane's worth, after the clock
        if use_dist:
            tt = torch.tensor([dt, cpu_s], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, cpu_s = float(tt[0].item()), float(tt[1].item())
        if rank == 0:
            out = dict(common)
            out.update({
                "metric": "proofs/sec for m=64 matmul circuit, batch of %d independent proofs from HOST witness matrices sharded over the GPUs (BASELINE configs[3], incl. H2D)" % args.batch,
                "value": args.batch / dt, "steps": args.batch, "warmup": nl, "ms_per_step": dt / args.batch * 1e3, "scaling": "strong",
                "config": {"workload": "prove_matmul_m%d_batch%d_host_witness" % (m, args.batch), "proof_bytes": len(res[0].to_bytes()),
                           "proofs_in_flight_per_gpu": nl, "entry_point": "gl_prove_columns (135 host vectors of n per proof, %.1f MB, pageable memory)" % (135 * hc.n * 8 / 1e6),
                           "includes": "H2D of every witness through the library's pinned ring, prove(), RCCL all_gather of the Merkle caps",
                           "sampled_proofs_verify": bool(ok), "host_cpu_seconds_per_proof_max_rank": cpu_s / max(1, len(mine)),
                           "parallelism": "proof i on rank i mod N; no data-path collective"},
                "roofline": None, "cpu_baseline": None})
            print(json.dumps(out))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.config4:
        dt, nbytes, ok = run_config4(args.batch, args.pool_lanes)
        if rank == 0:
            out = dict(common)
            out.update({
                "metric": "proofs/sec for m=64 matmul circuit, batch of %d independent random-witness proofs sharded over the GPUs (BASELINE configs[3])" % args.batch,
                "value": args.batch / dt, "steps": args.batch, "warmup": args.pool_lanes, "ms_per_step": dt / args.batch * 1e3, "scaling": "strong",
                "config": {"workload": "prove_matmul_m%d_batch%d" % (m, args.batch), "proof_bytes": nbytes, "proofs_in_flight_per_gpu": args.pool_lanes,
                           "includes": "operands from host memory, witness generation in HBM, prove(), RCCL all_gather of the Merkle caps",
                           "first_proof_verifies": bool(ok), "parallelism": "proof i on rank i mod N; no data-path collective"},
                "roofline": None, "cpu_baseline": None})
            print(json.dumps(out))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    cd = hc.build(ctx)                                    # circuit data replicated on every GPU
    # independent proofs overlap on separate streams (the transcript forces ~10 host syncs inside one proof)
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
    torch.cuda.synchronize()

    def step(i, lane=0):
        t, pis = wit[i % len(wit)]
        return lanes[lane][1].prove_device(ctypes.c_void_p(t.data_ptr()), pis)

    last_proof = [None] * nstreams                      # the last proof of every lane: verified AFTER the clock stops

    def run_steps(first, count, sink):
        """`count` proofs, round-robin over the lanes, one host thread per lane (ctypes releases the GIL)."""
        def work(lane):
            pr = None
            for i in range(first + lane, first + count, nstreams):
                pr = step(i, lane)
                sink[i - first] = pr.caps()
            last_proof[lane] = pr
            lanes[lane][0].synchronize()
        ths = [threading.Thread(target=work, args=(k,)) for k in range(nstreams)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    # a step = one batch of `nstreams` proofs (what a GPU keeps in flight); the lanes run continuously, not batch by batch
    per_step = nstreams
    nproofs = args.steps * per_step
    nwarm = max(1, args.warmup) * per_step
    run_steps(0, nwarm, [None] * nwarm)
    if use_dist:                                            # untimed: brings the RCCL communicator up
        sharding.gather_caps(np.zeros((1, 3, 16, 4), dtype=np.uint64), world, device=dev)
        if os.environ.get("BENCH_RCCL_DIAG") == "destroy" and world == 1:      # diagnosis only (profiles/README.md)
            dist.destroy_process_group()
            use_dist = False
    barrier()
    caps = [None] * nproofs
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    run_steps(0, nproofs, caps)
    cpu_timed = time.process_time() - cpu0
    gathered = None
    if use_dist:                                            # the Merkle-cap gather: 3 x 16 x 32 B per proof
        # rank r proved global proofs r, r + world, ... (round-robin); every rank ends with all caps in proof order
        gathered = sharding.gather_caps(np.stack(caps), world * nproofs, device=dev)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        assert gathered.shape == (world * nproofs, 3, 16, 4)
        assert (gathered[rank::world] == np.stack(caps)).all()

    proof_bytes = len(step(0).to_bytes())
    lanes_verified = sum(1 for pr in last_proof if pr is not None and cd.verify(pr)[0])      # one proof per lane out of the timed loop
    lanes_with_proofs = sum(1 for pr in last_proof if pr is not None)
    if rank == 0:
        value = world * nproofs / dt
        # secondary figure (not `value`): the same loop with witness generation inside the clock -- operands on the host,
        # arithmetic rows filled by the GPU, the sequential public-input hash sponge by the lane's host thread
        e2e = None
        if args.e2e_steps > 0 and world == 1:                # single-GPU runs only: the scaling runs measure `value`
            # the host part of witness generation (several ms of sequential Poseidon per proof) hides behind the other lanes' GPU work
            nl = args.e2e_lanes if args.e2e_lanes > 0 else nstreams
            el = lanes + [(lambda c: (c, p.api.CircuitView(cd, c)))(p.Context(device=local_rank)) for _ in range(nl - nstreams)]
            gens = [hc.witness_generator(c) for c, _ in el]
            bufs = [torch.empty((135, hc.n), dtype=torch.int64, device=dev) for _ in el]
            ops = [(np.random.default_rng(77 + k).integers(0, 2**32 - 1, m * m, dtype=np.uint64),
                    np.random.default_rng(177 + k).integers(0, 2**32 - 1, m * m, dtype=np.uint64)) for k in range(4)]
            torch.cuda.synchronize()

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
        # secondary figure (not `value`): the drop-in host path.  gl_prove_columns from full HOST witness matrices (135 pageable
        # vectors per proof, as plonk/prover.rs:145 holds them): 35 MB cross PCIe per proof inside the clock
        hostw = None
        if args.host_witness_steps > 0 and world == 1:
            hw_dt, hw_cpu, hw_res, _hw = host_witness_run(lanes, args.host_witness_steps, nstreams, 4, 900)
            t_same = torch.from_numpy(np.stack(_hw[0][0].cols).view(np.int64)).to(dev)
            torch.cuda.synchronize()
            same = hw_res[0].to_bytes() == lanes[0][1].prove_device(ctypes.c_void_p(t_same.data_ptr()), _hw[0][1]).to_bytes()
            hostw = {"value": args.host_witness_steps / hw_dt, "unit": "proofs/s", "proofs": args.host_witness_steps, "proofs_in_flight": nstreams,
                     "ms_per_proof": hw_dt / args.host_witness_steps * 1e3, "frac_of_resident_witness_rate": args.host_witness_steps / hw_dt / value,
                     "host_cpu_seconds_per_proof": hw_cpu / args.host_witness_steps, "h2d_bytes_per_proof": 135 * hc.n * 8,
                     "pcie_GBs": 135 * hc.n * 8 * args.host_witness_steps / hw_dt / 1e9, "bytes_equal_to_prove_device": bool(same),
                     "entry_point": "gl_prove_columns: 135 pageable host vectors per proof -> two-deep pinned ring -> HBM, then prove()"}
            del hw_res, _hw
        # the roofline kernel (forward 2^20 NTT x 64) and the Poseidon ceiling always run; --no-extra skips the other micro-kernel rows
        roofline, ntt, extra, perm_ceiling = ntt_leg(torch, ctx, lib, check, dev, args.ntt_batch, with_rows=not args.no_extra)
        ctx.timing(True)
        step(0)
        scopes = {k: round(v["ms"], 4) for k, v in ctx.timing_report().items()}
        ctx.timing(False)
        out = dict(common)
        out.update({
            "metric": "proofs/sec for m=64 matmul circuit (full prove(), PoseidonGoldilocksConfig, standard_recursion_config)",
            "value": value,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "scaling": "weak",
            "config": {"workload": "prove_matmul_m%d" % m, "trace_rows": hc.n, "lde_size": hc.n << 3, "proof_bytes": proof_bytes,
                       "step": "one batch of %d independent proofs per GPU" % per_step, "proofs_per_step_per_gpu": per_step,
                       "proofs_timed": world * nproofs, "ms_per_proof": dt / (world * nproofs) * 1e3,
                       "witnesses_per_gpu": len(wit), "proofs_in_flight_per_gpu": nstreams, "parallelism": "independent proofs per GPU; RCCL all_gather of Merkle caps only"},
            "roofline": roofline,
            "ntt": ntt,
            "extra": extra,
            "prove_device_ms_by_scope": scopes,
            "with_witness_generation": e2e,
            "with_host_witness": hostw,
        })
        out["config"]["lanes_verified_after_the_clock"] = "%d of %d" % (lanes_verified, lanes_with_proofs)
        out["host"]["cpu_seconds_timed_region_rank0"] = round(cpu_timed, 3)
        out["host"]["cpu_seconds_per_proof_rank0"] = cpu_timed / max(1, nproofs)
        if perm_ceiling:
            perms = permutations_per_proof(hc.n)
            per_gpu = value / world
            out["roofline_prove"] = {
                "bound": "integer ALU (Poseidon permutations); MFMA never: no dense contraction on this path",
                "permutations_per_proof": perms, "achieved_permutations_per_s": per_gpu * perms, "peak_permutations_per_s": perm_ceiling,
                "frac": per_gpu * perms / perm_ceiling,
                "peak_is": "measured on this box in this run: gl_poseidon_permute, one state per lane, 2^23 states",
                "hbm": {"algorithmic_bytes_per_proof": ALGO_BYTES_PER_PROOF_M64 if m == 64 else None,
                        "achieved_GBs": per_gpu * ALGO_BYTES_PER_PROOF_M64 / 1e9 if m == 64 else None,
                        "frac_of_peak": per_gpu * ALGO_BYTES_PER_PROOF_M64 / 1e9 / HBM_PEAK_GBS if m == 64 else None}}
            vp = os.path.join(ROOT, "profiles", "prove_m64_valu.json")
            if m == 64 and os.path.exists(vp):
                # the whole proof against the VALU issue roof: every kernel family of prove() is integer-ALU work, and a SIMD
                # issues one wave-wide VALU instruction per 4 cycles (16 lanes), whatever the opcode (profiles/README.md)
                try:
                    per_proof = json.load(open(vp))["valu_wave_instructions_per_proof"]
                    peak = 256 * 4 * VALU_CLOCK_HZ / 4.0
                    out["roofline_prove"]["valu_issue"] = {
                        "wave_instructions_per_proof": per_proof, "achieved_per_s": per_gpu * per_proof, "peak_per_s": peak,
                        "frac": per_gpu * per_proof / peak,
                        "peak_is": "256 CUs x 4 SIMDs x %.1f GHz / 4 cycles per wave64 instruction" % (VALU_CLOCK_HZ / 1e9),
                        "source": "profiles/prove_m64_valu.json: SQ_INSTS_VALU of tools/prove_profile.py, difference of two proof counts "
                                  "(committed profile, not measured in this run)"}
                except Exception:
                    pass
        if world == 1 and not args.no_extra:
            del lanes[1:]
            c4_dt, _, c4_ok = run_config4(args.batch, args.pool_lanes)
            out["config4_batch512"] = {"value": args.batch / c4_dt, "unit": "proofs/s", "proofs": args.batch, "seconds": round(c4_dt, 3), "n_gpus": 1,
                                       "proofs_in_flight": args.pool_lanes, "first_proof_verifies": bool(c4_ok),
                                       "includes": "operands from host memory, witness generation in HBM, prove(); the multi-GPU form is `bench.py --gpus N --config4`"}
        out["cpu_baseline"] = cpu_baseline(m, rows=args.cpu_rows, one_thread_full=args.cpu_one_thread) if (world == 1 and not args.no_cpu) else None
        if out["cpu_baseline"]:
            # BASELINE.md publishes no number for this metric: the ratio below is against the CPU PORT timed in this run (kind "port"),
            # a reported baseline and not a kernel-quality figure (the roofline fractions are)
            out["vs_baseline"] = value / out["cpu_baseline"]["value"]
            out["vs_baseline_is"] = "value / cpu_baseline.value (kind: port, %d threads)" % out["cpu_baseline"]["threads"]
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
