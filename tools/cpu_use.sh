#!/bin/bash
# host CPU seconds and cgroup throttling of one bench.py prove loop:  tools/cpu_use.sh [extra bench flags]
cd "$(dirname "$0")/.."
stat() { grep "^$1 " /sys/fs/cgroup/cpu.stat 2>/dev/null | cut -d' ' -f2; }
u0=$(stat usage_usec); t0=$(stat nr_throttled); s=$(date +%s.%N)
v=$(python3 bench.py --no-cpu --no-extra --e2e-steps 0 --steps 60 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f proofs/s, timed region %.2f s' % (d['value'], d['ms_per_step']*d['steps']/1e3))")
e=$(date +%s.%N); u1=$(stat usage_usec); t1=$(stat nr_throttled)
python3 -c "print('$v | process wall %.1f s, CPU %.1f s, throttled periods %d' % ($e - $s, (${u1:-0} - ${u0:-0}) / 1e6, ${t1:-0} - ${t0:-0}))"
