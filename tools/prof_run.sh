#!/bin/bash
# rocprofv3 kernel-trace + stats of one bench.py invocation on the GPU box.
# usage: tools/prof_run.sh <name> <bench args...>   -> gpurun_out/prof_<name>/ + summary json
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$REPO/gpurun_out/prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $REPO/bench.py "$@" > $OUT/bench.log 2>&1
grep "^{\"metric\"" $OUT/bench.log | tail -1 > $OUT/bench_line.json
JAMD_BY_GRID=1 python $REPO/tools/rocpd_summary.py $OUT "" > $OUT/summary.json 2>/dev/null
find $OUT -name "*.db" -delete
ls $OUT
