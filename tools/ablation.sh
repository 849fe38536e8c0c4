#!/bin/bash
# Memory-only time of the two NTT passes: builds a diagnostic copy of the library with -DNTT_ABLATION (the butterfly stages /
# loads / stores of ntt_col_pass and ntt_row_pass can then be switched off with GL_NTT_DEBUG = 1 | 2 | 4) and times the 2^20 x 64
# forward transform in each mode.  Run on the GPU box: bash tools/ablation.sh > gpurun_out/ntt_ablation.txt
set -e
cd "$(dirname "$0")/.."
CS=plonky2_demo_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -DNTT_ABLATION -c $CS/ntt.hip -o /tmp/ntt_abl.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libplonky2_mi355x_ablation.so /tmp/ntt_abl.o $(ls $CS/*.o | grep -v "/ntt.o")
export PLONKY2_MI355X_LIB=/tmp/libplonky2_mi355x_ablation.so
for mode in 0 6 2 4 1 3 5 7; do
  case $mode in
    0) what="full kernels";;
    6) what="butterfly stages only: no global loads, no global stores (the VALU phase)";;
    2) what="no global loads";;
    4) what="no global stores";;
    1) what="no butterfly stages (global load, LDS transpose, inter-pass twiddle multiply, global store)";;
    3) what="no stages, no global loads (LDS + twiddle multiply + stores)";;
    5) what="no stages, no global stores (loads + LDS + twiddle multiply)";;
    7) what="no stages, loads or stores (LDS traffic and the twiddle multiply only)";;
  esac
  echo "== GL_NTT_DEBUG=$mode: $what"
  GL_NTT_DEBUG=$mode python3 tools/time_ntt.py 20 64 40 2>&1 | grep -E "pass\(forward|^forward"
done
