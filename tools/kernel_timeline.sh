#!/bin/bash
# rocprofv3 kernel trace of one bench.py invocation -> start / end of every dispatch of the two big kernels (ms from the
# first one): who overlaps whom.   usage: tools/kernel_timeline.sh <name> <bench args...>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$REPO/gpurun_out/timeline_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python $REPO/bench.py "$@" > $OUT/bench.log 2>&1
python - <<PY > $OUT/timeline.txt
import sqlite3, glob
for f in glob.glob("$OUT/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    big = [(n, s, e) for n, s, e in rows if ("beam_exact" in n or "gmm_tile" in n)]
    if not big: continue
    t0 = big[0][1]
    for n, s, e in big:
        print(f"{(s - t0) / 1e6:10.2f} {(e - t0) / 1e6:10.2f}  {(e - s) / 1e6:8.2f} ms  {n[:60]}")
PY
find $OUT -name "*.db" -delete
tail -60 $OUT/timeline.txt
