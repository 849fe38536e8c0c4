#!/usr/bin/env python3
"""Per-thread host CPU time of a bench.py run (samples /proc/<pid>/task/*/stat): python tools/cpu_threads.py [bench flags]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tck = os.sysconf("SC_CLK_TCK")
proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra", "--e2e-steps", "0", "--steps", "60"] + sys.argv[1:],
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
seen, running, samples = {}, {}, 0
while proc.poll() is None:
    try:
        for tid in os.listdir("/proc/%d/task" % proc.pid):
            try:
                st = open("/proc/%d/task/%s/stat" % (proc.pid, tid)).read()
                comm = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                seen[tid] = (comm, (int(f[11]) + int(f[12])) / tck, int(f[11]) / tck, int(f[12]) / tck)
                if f[0] == "R":
                    running[tid] = running.get(tid, 0) + 1
            except (OSError, ValueError):
                pass
    except OSError:
        break
    samples += 1
    time.sleep(0.05)
out = proc.stdout.read()
try:
    d = json.loads(out.strip().splitlines()[-1]); print("%.1f proofs/s" % d["value"])
except Exception:
    print("no bench line")
tot = sum(v[1] for v in seen.values())
print("threads seen %d, total CPU %.1f s over %d samples" % (len(seen), tot, samples))
for tid, (comm, cpu, u, s) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  tid %-8s %-18s cpu %6.2f s (user %.2f sys %.2f)  running in %d samples" % (tid, comm, cpu, u, s, running.get(tid, 0)))
