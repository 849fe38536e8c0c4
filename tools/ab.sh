#!/bin/bash
# A/B of library builds / environment knobs / bench flags on ONE box, legs run in the order given (alternate them):
#   tools/ab.sh "<label>:<env assignments>[:<bench flags>]" ...      AB_STEPS=120 (1920 proofs) for +-0.3 % resolution (default 40 = 640 proofs: +-1.5 %)
cd "$(dirname "$0")/.."
for leg in "$@"; do
  IFS=':' read -r label envs flags <<< "$leg"
  v=$(env $envs python3 bench.py --no-cpu --no-extra --e2e-steps 0 --steps ${AB_STEPS:-40} $flags 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f proofs/s' % d['value'])")
  echo "$label: $v"
done
