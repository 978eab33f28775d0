#!/bin/bash
# Profile the GMM bench on the GPU box: kernel-trace stats + PMC passes (each in
# its own run; --pmc never combined with other trace domains).
# usage: tools/prof_gmm.sh <outdir-under-gpurun_out> [JAMD_GMM_VARIANT]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof}
VAR=${2:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export JAMD_GMM_VARIANT=$VAR
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
pass() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/pmc_$n -o pmc -- $BENCH > $OUT/pmc_$n.log 2>&1
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
