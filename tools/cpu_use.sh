#!/bin/bash
# host CPU seconds and cgroup throttling of one bench.py prove loop:  tools/cpu_use.sh [extra bench flags]
cd "$(dirname "$0")/.."
u0=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); t0=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
s=$(date +%s.%N)
v=$(python3 bench.py --no-cpu --no-extra --e2e-steps 0 --steps 960 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f proofs/s, timed region %.2f s' % (d['value'], d['ms_per_step']*d['steps']/1e3))")
e=$(date +%s.%N)
u1=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); t1=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
echo "$v | process wall $(echo "$e - $s" | bc) s, CPU $(echo "($u1 - $u0) / 1000000" | bc -l | cut -c1-6) s, throttled periods $((t1 - t0))"
