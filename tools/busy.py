#!/usr/bin/env python3
"""GPU busy fraction from a rocprofv3 --kernel-trace CSV: union of the kernel intervals over the window of the prove() loop, and the
time by kernel name.  python tools/busy.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
# the window of the prove() loop: from the k_quotient launch `skip` of the way through (warm-up and set-up dropped) to the last
# one -- the trace of bench.py also holds the NTT / Poseidon legs, which run alone with host gaps between their launches
q = [x[0] for x in iv if x[2].startswith("void k_quotient<false>") or x[2].startswith("k_quotient<false>")]
if len(q) >= 8:
    lo, hi = q[int(len(q) * skip)], q[-1 - max(3, len(q) // 50)]      # (the last few proofs are bench.py's single scoped ones, after the other legs)
    iv = [x for x in iv if lo <= x[0] <= hi]
else:
    t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip       # no proofs in the trace: drop the set-up phase
    iv = [x for x in iv if x[0] >= t0]
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
by = collections.Counter()
for s, e, n in iv:
    by[n.split("(")[0][:60]] += e - s
print("kernels %d  span %.3f ms  busy (union) %.3f ms = %.1f %%  sum of kernel durations %.3f ms (%.2fx overlap)" % (len(iv), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(by.values()) / 1e6, sum(by.values()) / busy))
for n, t in by.most_common(14):
    print("  %-62s %9.3f ms  %5.1f %%" % (n, t / 1e6, 100.0 * t / sum(by.values())))
