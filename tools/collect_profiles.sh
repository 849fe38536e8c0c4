#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box call:  bash tools/collect_profiles.sh r02
# (rocprofv3: kernel-trace/stats runs and PMC runs are separate invocations, the program directly after `--`).
set -e
R=${1:-rXX}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1] bench.py (default flags)"
python3 $ROOT/bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
echo "[2] kernel stats of the prove() loop"
rocprofv3 --kernel-trace --stats -d $OUT/kt_prove -o kt --output-format csv -- python3 $ROOT/bench.py --steps 20 --no-cpu --no-extra --e2e-steps 0 > /dev/null 2>&1
cp $OUT/kt_prove/kt_kernel_stats.csv $OUT/${R}_bench_prove_m64_kernel_stats.csv
python3 $ROOT/tools/busy.py $OUT/kt_prove/kt_kernel_trace.csv 0.5 > $OUT/${R}_bench_prove_m64_gpu_busy.txt
echo "[3] kernel stats of the 2^20 NTT x 64 ([2] holds them too: bench.py --no-extra keeps the roofline leg, 2^20 x 64 launches only)"
rocprofv3 --kernel-trace --stats -d $OUT/kt_ntt -o kt --output-format csv -- python3 $ROOT/tools/prof_ntt.py 64 40 > /dev/null 2>&1
cp $OUT/kt_ntt/kt_kernel_stats.csv $OUT/${R}_ntt20_kernel_stats.csv
echo "[4] HBM traffic of the forward NTT (separate FETCH_SIZE / WRITE_SIZE passes)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 $ROOT/tools/prof_traffic.py 64 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 $ROOT/tools/prof_traffic.py 64 3 > /dev/null 2>&1
python3 $ROOT/tools/traffic_report.py $OUT 64 $OUT/ntt20_traffic.json > /dev/null
cp $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $OUT/${R}_ntt20_pmc_FETCH_SIZE.csv
cp $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/${R}_ntt20_pmc_WRITE_SIZE.csv
echo "[5] SQ counters: NTT and prove() kernels"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq_ntt -o s --output-format csv -- python3 $ROOT/tools/prof_ntt.py 64 2 > /dev/null 2>&1
cp $(find $OUT/sq_ntt -name "*counter_collection.csv" | head -1) $OUT/${R}_ntt20_pmc_sq_counters.csv
python3 $ROOT/tools/ntt_valu_report.py $OUT/${R}_ntt20_pmc_sq_counters.csv 64 $OUT/ntt20_valu.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq_prove -o s --output-format csv -- python3 $ROOT/tools/prove_profile.py 64 2 > $OUT/${R}_prove_m64_scopes.txt 2>&1
cp $(find $OUT/sq_prove -name "*counter_collection.csv" | head -1) $OUT/${R}_prove_m64_pmc_sq_counters.csv
# the per-proof instruction count of the THROUGHPUT configuration (what bench.py's 16 proofs in flight run): the profile proves one
# at a time, so the two choices the library makes from the number of proofs in flight are pinned through their knobs
export GL_COOP_MAX_NODES=1024 GL_POW_WINDOW_LOG=0
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $OUT/sq_prove2t -o s --output-format csv -- python3 $ROOT/tools/prove_profile.py 64 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $OUT/sq_prove8 -o s --output-format csv -- python3 $ROOT/tools/prove_profile.py 64 8 > /dev/null 2>&1
python3 $ROOT/tools/valu_report.py $(find $OUT/sq_prove2t -name "*counter_collection.csv" | head -1) $(find $OUT/sq_prove8 -name "*counter_collection.csv" | head -1) $OUT/prove_m64_valu.json > /dev/null
unset GL_COOP_MAX_NODES GL_POW_WINDOW_LOG
echo "[6] NTT ablation (memory-only time of the passes)"
cd $ROOT && bash tools/ablation.sh > $OUT/${R}_ntt_ablation.txt 2>&1
echo "[7] m = 128 (config 5): scopes, kernel stats, HBM traffic per kernel (separate passes)"
python3 $ROOT/tools/prove_profile.py 128 3 > $OUT/${R}_prove_m128_scopes.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_m128 -o kt --output-format csv -- python3 $ROOT/tools/prove_profile.py 128 3 > /dev/null 2>&1
cp $OUT/kt_m128/kt_kernel_stats.csv $OUT/${R}_prove_m128_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch128 -o f --output-format csv -- python3 $ROOT/tools/prove_profile.py 128 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write128 -o w --output-format csv -- python3 $ROOT/tools/prove_profile.py 128 1 > /dev/null 2>&1
cp $(find $OUT/pmc_fetch128 -name "*counter_collection.csv" | head -1) $OUT/${R}_m128_pmc_FETCH_SIZE.csv
cp $(find $OUT/pmc_write128 -name "*counter_collection.csv" | head -1) $OUT/${R}_m128_pmc_WRITE_SIZE.csv
CORR=$(python3 -c "import json; print(json.load(open('$OUT/ntt20_traffic.json'))['calibration']['read_correction'])")
python3 $ROOT/tools/kernel_traffic_report.py $OUT/${R}_prove_m128_kernel_stats.csv $OUT/${R}_m128_pmc_FETCH_SIZE.csv $OUT/${R}_m128_pmc_WRITE_SIZE.csv $CORR $OUT/${R}_m128_traffic.json > $OUT/${R}_m128_traffic.md
echo "[8] primitives: edge-grid check + cycle counts, SGPR hazard probe, clock probe"
cd $ROOT
tools/ubench/bin/gl_prims > $OUT/${R}_gl_primitives.txt 2>&1 || echo "gl_prims FAILED" >> $OUT/${R}_gl_primitives.txt
tools/ubench/bin/sgpr_hazard > $OUT/${R}_sgpr_hazard_probe.txt 2>&1
tools/ubench/bin/clock_probe > $OUT/${R}_clock_probe.txt 2>&1
rm -rf $OUT/kt_prove $OUT/kt_ntt $OUT/pmc_fetch $OUT/pmc_write $OUT/sq_ntt $OUT/sq_prove $OUT/sq_prove2t $OUT/sq_prove8 $OUT/kt_m128 $OUT/pmc_fetch128 $OUT/pmc_write128
ls -la $OUT
