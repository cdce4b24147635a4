#!/bin/bash
# ON the GPU box: run the given steps one after another, each under its own `timeout -k 10`.  An ordinary failure (a red test,
# a Python exception) is logged and the chain goes on; a step that was KILLED (timeout 124 / 137, a signal >= 128) ends the
# chain: no further GPU step is started behind a hang or a fault.
#   tools/gpu_chain.sh "name|seconds|command" ...        logs: gpurun_out/<name>.log, summary: gpurun_out/chain.txt
mkdir -p gpurun_out
: > gpurun_out/chain.txt
for step in "$@"; do
    name="${step%%|*}"; rest="${step#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "[chain] $name (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "$name rc=$rc $(( $(date +%s) - start ))s" | tee -a gpurun_out/chain.txt
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -ge 124 ]; then echo "[chain] $name was killed (rc $rc): stopping" | tee -a gpurun_out/chain.txt; exit $rc; fi
done
exit 0
