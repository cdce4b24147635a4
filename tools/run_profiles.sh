#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 passes of bench.py whose summaries tools/make_profile_summary.py turns into profiles/NAME_*.
#   tools/run_profiles.sh NAME [extra bench.py flags]
# One --kernel-trace --stats pass, then the three PMC passes on their own (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --side-steps 0 --c5-n 0 --c2-iters 0 --check-steps 0 --grid-steps 0 --resnext-steps 0 --no-certify $*"   # the timed steps only: every layer launch carries 512 clips
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$NAME -- $BENCH --steps 5 --warmup 1 > $OUT/prof_$NAME.json 2> $OUT/prof_$NAME.err || exit 1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass; tag=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${NAME}_$tag -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_${NAME}_$tag.log 2>&1 || exit 1
done
tail -1 $OUT/prof_$NAME.json | cut -c1-300
