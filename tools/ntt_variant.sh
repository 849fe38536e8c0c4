#!/bin/bash
# Builds a variant of the library whose ntt.hip is compiled with extra flags (the other objects are the product's), for A/B
# timing on the GPU box through PLONKY2_MI355X_LIB:   bash tools/ntt_variant.sh NAME -DNTT_TILE_LOG=12 -DNTT_THREADS=256 ...
# -> gpurun_out/variants/libplonky2_mi355x_NAME.so   (gpurun_out/ is scratch; the .so travels to the box with the snapshot? no:
#    gpurun_out/ is NOT sent, so variants are written under tools/variants/, which is git-ignored)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
CS=plonky2_demo_amd/csrc
mkdir -p tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off "$@" -c $CS/ntt.hip -o tools/variants/ntt_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libplonky2_mi355x_$name.so tools/variants/ntt_$name.o $(ls $CS/*.o | grep -v "/ntt.o")
echo tools/variants/libplonky2_mi355x_$name.so
