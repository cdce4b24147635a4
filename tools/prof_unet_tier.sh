#!/bin/bash
# ON the GPU box: rocprofv3 kernel trace + stats of one UNet evaluation of B spectrograms on tier TIER (0 fp32, 1 16-bit, 2 split-f16).
#   tools/prof_unet_tier.sh NAME TIER B      -> gpurun_out/prof_NAME/ (csv), gpurun_out/prof_NAME.txt (per-launch table of the conv GEMMs)
set -u
NAME=$1; TIER=$2; B=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TIER B
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -- python3 $ROOT/tools/gpu_unet_layers.py > $OUT/prof_$NAME.log 2>&1 || { tail -5 $OUT/prof_$NAME.log; exit 1; }
KERNEL=$([ "$TIER" = 2 ] && echo gemm_x3 || { [ "$TIER" = 1 ] && echo gemm_h16 || echo gemm_f32_kernel; }) ALL=1 python3 $ROOT/tools/gpu_unet_layers.py --analyse $OUT/prof_$NAME > $OUT/prof_$NAME.txt 2>&1
tail -3 $OUT/prof_$NAME.txt
f=$(find $OUT/prof_$NAME -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
python3 $ROOT/tools/gpu_unet_layers.py --time | tail -1
