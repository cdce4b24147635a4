#!/bin/bash
# Ablation of the split-f16 tier's kernel (gemm_x3_kernel, csrc/gemm_f32.hip): which part of a k-step pair costs what.
#   tools/x3_ablation.sh build      HERE (hipcc cross-compiles): libdmad_hip.so.<V> next to the product library, one per variant
#   tools/x3_ablation.sh run        ON the GPU box (gpurun): per variant a kernel trace of one evaluation (tools/x3_trace_split.py),
#                                   board power / sclk under back-to-back evaluations (tools/gpu_tier2_power.py) and, for the
#                                   stamped builds, in-kernel cycles per pair of the K = 9216 skip GEMM (device printf)
# The variants switch parts of the main loop OFF (results are then numerically meaningless; only time and power are read):
#   full      the product kernel                        nodma     no steady-state LDS-DMA (-DX3_NO_DMA)
#   nolds     no fragment ds_reads (-DX3_NO_LDS)         nofix     no hi/lo register exchange (-DX3_NO_FIX)
#   mfma      MFMAs + exchange only                      mfma0     MFMAs only
#   onlyA / onlyX   steady-state DMA of the weight / the activation pieces only
#   novmwait  DMA issued, its counted vmcnt wait removed
#   stamps, stamps_nodma, stamps_nolds   s_memtime per phase group (wave 0 / 3 / 4 of two workgroups)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/diffusion-model-for-audio-defense_amd
VARIANTS="full: nodma:-DX3_NO_DMA nolds:-DX3_NO_LDS nofix:-DX3_NO_FIX mfma:-DX3_NO_DMA,-DX3_NO_BARRIER,-DX3_NO_LDS mfma0:-DX3_NO_DMA,-DX3_NO_BARRIER,-DX3_NO_LDS,-DX3_NO_FIX onlyA:-DX3_ONLY_A onlyX:-DX3_ONLY_X novmwait:-DX3_NO_VMWAIT stamps:-DX3_STAMPS stamps_nodma:-DX3_STAMPS,-DX3_NO_DMA stamps_nolds:-DX3_STAMPS,-DX3_NO_LDS"
case "${1:-}" in
build)
    make -C $PKG/csrc >/dev/null || exit 1
    for v in $VARIANTS; do
        name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $PKG/csrc/gemm_f32.hip -o /tmp/gemm_f32_$name.o || exit 1
        objs=$(ls $PKG/csrc/*.o | grep -v gemm_f32.o)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libdmad_hip.so.$name $objs /tmp/gemm_f32_$name.o || exit 1
        echo "built libdmad_hip.so.$name ($flags)"
    done ;;
run)
    cd /tmp && export TMPDIR=/tmp && cd $ROOT
    OUT=gpurun_out/x3_ablation; mkdir -p $OUT; : > $OUT/summary.txt
    for v in $VARIANTS; do
        name=${v%%:*}; lib=$PKG/libdmad_hip.so.$name
        case $name in
        stamps*) DMAD_LIB=$lib B=19 PATHS=2 timeout -k 10 120 python3 tools/gpu_tier_time.py 2>&1 | grep -E 'x3 stamps' | sort | uniq | tail -6 | sed "s/^/$name /" >> $OUT/summary.txt || exit 1 ;;
        *)  DMAD_LIB=$lib B=19 PATHS=2 timeout -k 10 120 rocprofv3 --kernel-trace -d $OUT/trace_$name -o t -- python3 tools/gpu_tier_time.py > $OUT/trace_$name.log 2>&1 || exit 1
            python3 tools/x3_trace_split.py $OUT/trace_$name/t_results.db | sed "s/^/$name /" >> $OUT/summary.txt
            DMAD_LIB=$lib B=19 SECONDS=3 PATH_ID=2 timeout -k 10 120 python3 tools/gpu_tier2_power.py 2>/dev/null | tail -1 | sed "s/^/$name /" >> $OUT/summary.txt || exit 1 ;;
        esac
    done
    cat $OUT/summary.txt ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
