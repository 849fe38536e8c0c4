for t in 0 1024 2048 4096 8192 16384; do
  echo "coop threshold $t"
  GL_COOP_MAX_NODES=$t timeout -k 10 200 python bench.py --no-cpu --e2e-steps 0 --steps 64 > gpurun_out/bs.log 2>&1
  python3 -c "
import json;d=json.loads(open('gpurun_out/bs.log').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
done
