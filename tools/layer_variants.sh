#!/bin/bash
# Gate-phase variants of the dominant kernel (wn_layer_p, csrc/wn_layer.hip: WNL_VARIANT): what does the gate cost, and can it cost less?
#   tools/layer_variants.sh build     HERE (hipcc cross-compiles): libdmad_hip.so.v<N> next to the product library
#   tools/layer_variants.sh run       ON the GPU box: per variant the per-phase cycle stamps (DMAD_LAYER_STAMPS=1), the un-stamped ms per
#                                     launch and board power / in-kernel clock under back-to-back launches (tools/gpu_power_trace.py)
# Variants: 0 product (packed fp32 gate math) · 1 the gate's FMA steps as single v_add / v_fma (scalar code, -fno-slp-vectorize) ·
#           2 ABLATION: no transcendentals (numerically meaningless) · 3 GEMM2's MFMAs interleaved by sched_group_barrier ·
#           4 ABLATION: no epilogue (no residual add, no h' stores; tools/patches/wnl_variant4_no_epilogue.patch; VARS="0 4")
#           -> gpurun_out/layer_variants/summary.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/diffusion-model-for-audio-defense_amd
VARS="${VARS:-0 1 2 3}"
case "${1:-}" in
build)
    make -C $PKG/csrc >/dev/null || exit 1
    for v in $VARS; do
        src=$PKG/csrc/wn_layer.hip
        if [ $v -ge 4 ]; then          # variants 4-8 live in patches (the product source stays under bench.py's hash guard)
            pf=$ROOT/tools/patches/wnl_variant4_no_epilogue.patch; [ $v -ge 5 ] && pf=$ROOT/tools/patches/wnl_variant56_idle_cycles.patch; [ $v -ge 7 ] && pf=$ROOT/tools/patches/wnl_variant78_cu_skew.patch
            rm -rf /tmp/wnl_v$v && mkdir -p /tmp/wnl_v$v && cp $PKG/csrc/*.h $PKG/csrc/wn_layer.hip /tmp/wnl_v$v/ || exit 1
            (cd /tmp/wnl_v$v && patch -s -p3 < $pf) || exit 1
            src=/tmp/wnl_v$v/wn_layer.hip
        fi
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $([ $v = 1 ] && echo -fno-slp-vectorize) ${EXTRA:-} -DWNL_VARIANT=$v -c $src -o /tmp/wn_layer_v$v.o || exit 1
        objs=$(ls $PKG/csrc/*.o | grep -v wn_layer.o)
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libdmad_hip.so.v$v $objs /tmp/wn_layer_v$v.o || exit 1
        echo "built libdmad_hip.so.v$v"
    done ;;
run)
    OUT=gpurun_out/layer_variants; mkdir -p $OUT; : > $OUT/summary.txt
    for v in $VARS; do
        lib=$PKG/libdmad_hip.so.v$v
        echo "== variant $v" | tee -a $OUT/summary.txt
        DMAD_LIB=$lib B=${B:-256} SECONDS=4 timeout -k 10 240 python3 tools/gpu_power_trace.py > $OUT/power_v$v.log 2>&1 || { tail -5 $OUT/power_v$v.log; exit 1; }
        grep -E "^random|^zeros|dmad stamps" $OUT/power_v$v.log | cut -c1-420 | tee -a $OUT/summary.txt
        cp gpurun_out/power_trace.json $OUT/power_trace_v$v.json 2>/dev/null
    done ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
