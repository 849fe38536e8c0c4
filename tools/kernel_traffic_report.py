#!/usr/bin/env python3
"""Per-kernel HBM traffic and rate of one profiled program (BASELINE configs[4]: m = 128):
  python tools/kernel_traffic_report.py <kernel_stats.csv> <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> <read correction> <out json> [proofs]
FETCH_SIZE / WRITE_SIZE are KiB per dispatch (MI355X_MICROARCH.md, HBM section; separate --pmc passes); the read side is multiplied
by the correction calibrated on a kernel of known bytes in the same session (tools/traffic_report.py: ~2.0, the guide's "FETCH_SIZE
reports half" rule).  Durations come from the --kernel-trace --stats run (no counters active)."""
import collections, csv, json, sys
stats, fetch, write, corr, out = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
proofs = int(sys.argv[6]) if len(sys.argv) > 6 else 1
PEAK = 8000.0

def short(k):
    return k.replace("void ", "").split("(")[0]

def counters(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc

f, w = counters(fetch, "FETCH_SIZE"), counters(write, "WRITE_SIZE")
rows = []
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    if k not in f:
        continue
    calls, avg_ns = int(r["Calls"]), float(r["AverageNs"])
    rd = sum(f[k]) / len(f[k]) * 1024.0 * corr
    wr = sum(w[k]) / len(w[k]) * 1024.0 if k in w else 0.0
    rows.append({"kernel": k, "launches": calls, "avg_us": avg_ns / 1e3, "total_ms": calls * avg_ns / 1e6, "hbm_read_MB_per_launch": rd / 1e6,
                 "hbm_write_MB_per_launch": wr / 1e6, "hbm_GBs": (rd + wr) / avg_ns, "frac_of_8TBs": (rd + wr) / avg_ns / PEAK})
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"read_correction": corr, "proofs_profiled": proofs, "kernels": rows,
           "source": "rocprofv3 --kernel-trace --stats and two --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same program"}, open(out, "w"), indent=1)
print("| kernel | launches | avg us | total ms | HBM read MB | HBM write MB | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|")
for r in rows[:24]:
    print("| `%s` | %d | %.1f | %.2f | %.1f | %.1f | %.0f | %.1f %% |" % (r["kernel"][:60], r["launches"], r["avg_us"], r["total_ms"], r["hbm_read_MB_per_launch"],
                                                                       r["hbm_write_MB_per_launch"], r["hbm_GBs"], 100 * r["frac_of_8TBs"]))
