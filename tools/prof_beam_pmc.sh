#!/bin/bash
# PMC passes for the first-pass kernel (one counter set per run).  usage: tools/prof_beam_pmc.sh <name> [utts]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; UTTS=${2:-256}
OUT=$REPO/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload e2e --utts $UTTS --no-cpu-baseline --no-batch --steps 1 --warmup 1"
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $OUT/$n -o pmc -- $BENCH > $OUT/$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
python $REPO/tools/rocpd_summary.py $OUT "beam_" > $OUT/summary.json 2>/dev/null
find $OUT -name "*.db" -delete
