#!/bin/bash
# A/B of whole-library compile flags (round 5: -fno-slp-vectorize — hipcc's SLP vectoriser packs adjacent fp32 adds / multiplies into
# v_pk_*_f32, which cost a power-capped board clock without saving cycles beside MFMAs).
#   tools/flag_ab.sh build NAME "FLAGS"    HERE: every csrc/*.hip with FLAGS -> libdmad_hip.so.NAME
#   tools/flag_ab.sh run NAME...            ON the GPU box: per library (and `product`) the layer kernel under the power sampler, the
#                                           tail kernel, one UNet evaluation per tier, the WaveNet's fp32 / split-f16 tiers
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/diffusion-model-for-audio-defense_amd
case "${1:-}" in
build)
    name=$2; flags=$3; objs=""
    for f in $PKG/csrc/*.hip; do
        b=$(basename $f .hip)
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $f -o /tmp/ab_${name}_$b.o || exit 1
        objs="$objs /tmp/ab_${name}_$b.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libdmad_hip.so.$name $objs || exit 1
    echo "built libdmad_hip.so.$name ($flags)" ;;
run)
    shift
    OUT=gpurun_out/flag_ab; mkdir -p $OUT; : > $OUT/summary.txt
    for name in product "$@"; do
        if [ $name = product ]; then unset DMAD_LIB; else export DMAD_LIB=$PKG/libdmad_hip.so.$name; fi
        echo "== $name" | tee -a $OUT/summary.txt
        B=256 SECONDS=3 timeout -k 10 240 python3 tools/gpu_power_trace.py > $OUT/power_$name.log 2>&1 || { tail -5 $OUT/power_$name.log; exit 1; }
        grep -E "^random" $OUT/power_$name.log | cut -c1-130 | tee -a $OUT/summary.txt
        grep -E "dmad stamps" $OUT/power_$name.log | head -1 | cut -c1-200 | tee -a $OUT/summary.txt
        B=512 SECONDS=3 timeout -k 10 240 python3 tools/gpu_final_time.py 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt || exit 1
        for t in 1 2 0; do
            TIER=$t B=$([ $t = 0 ] && echo 512 || echo 2048) REPS=10 timeout -k 10 200 python3 tools/gpu_unet_layers.py --time 2>&1 | tail -1 | sed "s/^/UNet tier $t: /" | cut -c1-200 | tee -a $OUT/summary.txt || exit 1
        done
        B=32 PATHS=2,1 timeout -k 10 200 python3 tools/gpu_tier_time.py 2>&1 | tail -2 | cut -c1-200 | tee -a $OUT/summary.txt || exit 1
    done ;;
*) echo "usage: $0 build NAME FLAGS | run NAME..."; exit 2 ;;
esac
