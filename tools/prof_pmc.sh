#!/bin/bash
# PMC passes for one bench.py invocation (each counter set in its own run; --pmc is
# never combined with trace domains).  usage: tools/prof_pmc.sh <name> <kernel-substr> <bench args...>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; SUB=$2; shift 2
OUT=$REPO/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $OUT/$n -o pmc -- python $REPO/bench.py "${ARGS[@]}" > $OUT/$n.log 2>&1; }
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY
pass mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
python $REPO/tools/rocpd_summary.py $OUT "$SUB" > $OUT/summary.json 2>/dev/null
find $OUT -name "*.db" -delete
