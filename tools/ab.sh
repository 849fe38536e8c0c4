#!/bin/bash
# A/B of library builds / environment knobs on ONE box: tools/ab.sh "<label>:<env assignments>" ...   (each leg: bench.py prove loop only)
cd "$(dirname "$0")/.."
for leg in "$@"; do
  label="${leg%%:*}"; envs="${leg#*:}"
  v=$(env $envs python3 bench.py --no-cpu --no-extra --e2e-steps 0 --steps ${AB_STEPS:-640} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f proofs/s' % d['value'])")
  echo "$label: $v"
done
