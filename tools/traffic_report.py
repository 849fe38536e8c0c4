#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/prof_traffic.py into profiles/ntt20_traffic.json.
Counter units are KiB per dispatch (MI355X_MICROARCH.md, HBM section); the read side is calibrated on the k_field_op
dispatches (known 8n bytes read with the same 8-byte-per-lane access shape) instead of assuming the x2 rule."""
import csv, glob, json, os, sys
root, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3          # forward NTTs prof_traffic.py ran (its second argument)
def load(pattern, counter):
    rows = []
    for f in glob.glob(os.path.join(root, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows
fetch, write = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
def avg(rows, key):
    v = [x for k, x in rows if key in k]
    return sum(v) / len(v) if v else None
n = B << 20
cal_f, cal_w = avg(fetch, "k_field_op"), avg(write, "k_field_op")
corr_f = (8.0 * n / 1024.0) / cal_f          # bytes-known / counter
corr_w = (8.0 * n / 1024.0) / cal_w
out = {"batch": B, "calibration": {"kernel": "k_field_op (8n B read, 8n B written)", "FETCH_SIZE_KiB": cal_f, "WRITE_SIZE_KiB": cal_w,
                                   "read_correction": corr_f, "write_correction": corr_w}}
per = {}
tot = 0.0
launches = {}
for key in ("ntt_col_pass", "ntt_row_pass"):
    f, w = avg(fetch, key), avg(write, key)
    cnt = len([1 for k, _ in fetch if key in k])
    per[key] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_read_bytes": f * 1024 * corr_f, "hbm_write_bytes": w * 1024 * corr_w, "dispatches_seen": cnt}
    launches[key] = cnt
out["per_launch"] = per
# one forward NTT over B polynomials = (B * 2^20 / inter-pass scratch elements) chunks x (col + row): counted from the dispatches seen
chunks = max(1, round(launches["ntt_col_pass"] / reps))
out["launches_per_forward_ntt"] = {"ntt_col_pass": chunks, "ntt_row_pass": chunks}
out["forward_ntt_hbm_bytes"] = chunks * sum(per[k]["hbm_read_bytes"] + per[k]["hbm_write_bytes"] for k in per)
out["algorithmic_bytes"] = 16.0 * n
out["traffic_over_algorithmic"] = out["forward_ntt_hbm_bytes"] / out["algorithmic_bytes"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
