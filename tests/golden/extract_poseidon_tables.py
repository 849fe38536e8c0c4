#!/usr/bin/env python3
"""Extracts the reference's precomputed Poseidon "fast partial round" tables as DATA (known-answer material, like the KATs):
plonky2/src/hash/poseidon_goldilocks.rs:24-215 -> tests/golden/poseidon_fast_tables.json.

Run in the build container only (it reads /root/reference as text); the JSON fixture is what travels.  The tables are the
values tools/gen_poseidon_constants.py must reproduce from the 360 round constants + the MDS definition, and the values the
PoseidonGate constraints (gates/poseidon.rs:193-272) are built from."""
import json
import os
import re
import sys

SRC = "/root/reference/plonky2/src/hash/poseidon_goldilocks.rs"
NAMES = ["MDS_MATRIX_CIRC", "MDS_MATRIX_DIAG", "FAST_PARTIAL_FIRST_ROUND_CONSTANT", "FAST_PARTIAL_ROUND_CONSTANTS",
         "FAST_PARTIAL_ROUND_VS", "FAST_PARTIAL_ROUND_W_HATS", "FAST_PARTIAL_ROUND_INITIAL_MATRIX"]


def main():
    text = open(SRC).read()
    out = {"source": "plonky2/src/hash/poseidon_goldilocks.rs:24-215"}
    for name in NAMES:
        m = re.search(r"const %s\s*:[^=]*=\s*\[(.*?)\];" % name, text, re.S)
        if not m:
            sys.exit("table %s not found" % name)
        vals = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", m.group(1))]
        out[name] = vals
    assert len(out["FAST_PARTIAL_FIRST_ROUND_CONSTANT"]) == 12 and len(out["FAST_PARTIAL_ROUND_CONSTANTS"]) == 22
    assert len(out["FAST_PARTIAL_ROUND_VS"]) == 22 * 11 and len(out["FAST_PARTIAL_ROUND_W_HATS"]) == 22 * 11
    assert len(out["FAST_PARTIAL_ROUND_INITIAL_MATRIX"]) == 11 * 11
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poseidon_fast_tables.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", dst)


if __name__ == "__main__":
    main()
