#!/bin/bash
# PMC passes for the DNN layer kernels (one counter set per run).  usage: tools/prof_dnn_pmc.sh <name> [variant]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; export JAMD_DNN_VARIANT=${2:-4}
OUT=$REPO/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload dnn --no-cpu-baseline --steps 2 --warmup 1"
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d $OUT/$n -o pmc -- $BENCH > $OUT/$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
python $REPO/tools/rocpd_summary.py $OUT "dnn_layer" > $OUT/summary.json 2>/dev/null
find $OUT -name "*.db" -delete
