#!/usr/bin/env python3
"""VALU instructions per m = 64 proof from two SQ-counter runs of tools/prove_profile.py with different proof counts (the
difference removes the circuit build):  python tools/valu_report.py <short counter csv> <long counter csv> <out json>
Counts are SQ_INSTS_VALU (wave-level instructions); one proof = one k_quotient<false> launch."""
import collections, csv, json, sys

def load(path):
    acc = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in launches.items()}

a, la = load(sys.argv[1])
b, lb = load(sys.argv[2])
proofs = lb["k_quotient<false>"] - la["k_quotient<false>"]
per = {k: (b[k] - a.get(k, 0.0)) / proofs for k in b if abs(b[k] - a.get(k, 0.0)) > 0}
total = sum(per.values())
poseidon = sum(v for k, v in per.items() if "merkle" in k or "pow_grind" in k)
out = {"proofs_in_difference": proofs, "valu_wave_instructions_per_proof": total,
       "share": {k: round(v / total, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v / total >= 0.002},
       "launches_per_proof": {k: round((lb[k] - la.get(k, 0)) / proofs, 2) for k in per if (lb[k] - la.get(k, 0))},
       "hash_kernels_share": round(poseidon / total, 4),
       "source": "rocprofv3 --pmc SQ_INSTS_VALU ... of tools/prove_profile.py 64 2 and 64 8 (tools/collect_profiles.sh), difference of the two runs, with the choices the library makes for several proofs in flight pinned (GL_COOP_MAX_NODES=1024 GL_POW_WINDOW_LOG=0)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
