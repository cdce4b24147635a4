#!/bin/bash
# Development (run ON THE GPU BOX): kernel-trace one UNet evaluation (B spectrograms) per library variant and print the conv GEMM totals.
#   tools/ab_trace.sh VARIANT...     VARIANT = none | pers (DMAD_H16_PERS=0: no persistent form) | sr0 (DMAD_H16_SR=0) | a suffix of libdmad_hip.so.<suffix> built beside the in-tree library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    unset DMAD_LIB DMAD_H16_PERS DMAD_H16_SR
    if [ $v = pers ]; then export DMAD_H16_PERS=0; elif [ $v = sr0 ]; then export DMAD_H16_SR=0; elif [ $v != none ]; then export DMAD_LIB=$R/diffusion-model-for-audio-defense_amd/libdmad_hip.so.$v; fi
    B=${B:-2048} timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/tools/gpu_unet_layers.py > /dev/null 2>&1 || exit 1
    echo "== $v"
    B=${B:-2048} KERNEL=gemm_h16 ALL=1 python3 $R/tools/gpu_unet_layers.py --analyse $R/gpurun_out/ab_$v > $R/gpurun_out/ab_$v.txt
    grep -E "output_blocks.9.0 conv1|input_blocks.7.0 conv2|input_blocks.2.0 conv2|output_blocks.13.0 conv1|input_blocks.7.1 qkv|total gemm" $R/gpurun_out/ab_$v.txt | grep -E "kernel|total" | sort -u
done
