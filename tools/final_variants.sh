#!/bin/bash
# What bounds wn_final_p (csrc/wn_final.hip)?  Ablation builds (WNF_VARIANT) timed under the board-power sampler.
#   tools/final_variants.sh build     HERE: libdmad_hip.so.f<N>          tools/final_variants.sh run     ON the GPU box
# Variants: 0 product · 1 gate rows from L2 (no HBM stream) · 2 no MFMAs in the skip GEMM (pure streamer)   -> gpurun_out/final_variants/summary.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/diffusion-model-for-audio-defense_amd
VARS="${VARS:-0 1 2}"
case "${1:-}" in
build)
    make -C $PKG/csrc >/dev/null || exit 1
    for v in $VARS; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DWNF_VARIANT=$v -c $PKG/csrc/wn_final.hip -o /tmp/wn_final_f$v.o || exit 1
        objs=$(ls $PKG/csrc/*.o | grep -v wn_final.o)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libdmad_hip.so.f$v $objs /tmp/wn_final_f$v.o || exit 1
        echo "built libdmad_hip.so.f$v"
    done ;;
run)
    OUT=gpurun_out/final_variants; mkdir -p $OUT; : > $OUT/summary.txt
    for v in $VARS; do
        echo "== variant $v" | tee -a $OUT/summary.txt
        DMAD_LIB=$PKG/libdmad_hip.so.f$v B=${B:-512} SECONDS=4 timeout -k 10 240 python3 tools/gpu_final_time.py 2>&1 | tail -2 | cut -c1-600 | tee -a $OUT/summary.txt || exit 1
    done ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
