#!/usr/bin/env python3
"""VALU instructions per element and wait shares of the 2^20 NTT passes from the SQ-counter pass of tools/prof_ntt.py:
  python tools/ntt_valu_report.py <counter_collection.csv> <batch> <out json>        (-> profiles/ntt20_valu.json, read by bench.py)"""
import collections, csv, json, sys
path, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    ps = "col" if "ntt_col_pass<10, false" in k else "row" if "ntt_row_pass<10, false" in k else None
    if ps:
        acc[ps][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[ps].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
elems = batch << 20
res = {"batch": batch, "valu_instructions_per_element": {}, "wait_any_share_of_wave_cycles": {}, "wait_inst_any_share_of_wave_cycles": {},
       "valu_active_share_of_wave_cycles": {}, "lds_instructions_per_element": {}, "salu_instructions_per_element": {}}
for ps, c in acc.items():
    mean = {k: sum(v) / len(v) for k, v in c.items()}
    res["valu_instructions_per_element"][ps] = mean["SQ_INSTS_VALU"] * 64.0 / elems
    res["lds_instructions_per_element"][ps] = mean.get("SQ_INSTS_LDS", 0.0) * 64.0 / elems
    res["salu_instructions_per_element"][ps] = mean.get("SQ_INSTS_SALU", 0.0) * 64.0 / elems
    res["wait_any_share_of_wave_cycles"][ps] = mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"]
    res["wait_inst_any_share_of_wave_cycles"][ps] = mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"]
    res["valu_active_share_of_wave_cycles"][ps] = mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"]
# The clock the kernels really ran at, and the share of it the vector ALUs were issuing: SQ_BUSY_CYCLES is summed over the 32
# shader engines and counts cycles, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES are summed over the 1024 SIMDs and count quad-cycles (the
# resulting average residency, 3.8 of the 4 waves per SIMD the kernels can hold, confirms both divisors)
res["clock_GHz"], res["valu_utilisation_at_that_clock"], res["resident_waves_per_simd"], res["duration_us_under_counters"] = {}, {}, {}, {}
for ps, c in acc.items():
    mean = {k: sum(v) / len(v) for k, v in c.items()}
    d_ns = sum(dur[ps]) / len(dur[ps])
    busy = mean["SQ_BUSY_CYCLES"] / 32.0
    res["duration_us_under_counters"][ps] = d_ns / 1e3
    res["clock_GHz"][ps] = busy / d_ns
    res["valu_utilisation_at_that_clock"][ps] = mean["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / busy
    res["resident_waves_per_simd"][ps] = mean["SQ_WAVE_CYCLES"] * 4.0 / 1024.0 / busy
res["source"] = "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- python3 tools/prof_ntt.py %d 2 (forward launches)" % batch
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
