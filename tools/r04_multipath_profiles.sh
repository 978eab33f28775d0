#!/bin/bash
# Round-4 rocprofv3 evidence for the multipath frame (csrc/beam_exact_mp.h): kernel trace next to the event-timed line of
# the same process, and two PMC passes (their own runs), on the C3 task decoded with -multipath.
set -u
R=${1:-r04}
O=gpurun_out/$R; mkdir -p $O
bash tools/prof_run.sh ${R}_e2e_mp --workload e2e --multipath --utts 256 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
cp gpurun_out/prof_${R}_e2e_mp/summary.json $O/e2e_mp_kernel_trace_summary.json 2>/dev/null; cp gpurun_out/prof_${R}_e2e_mp/bench_line.json $O/e2e_mp_bench_line_under_rocprof.json 2>/dev/null
( REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_${R}_beam_mp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
  BENCH="python $REPO/bench.py --workload e2e --multipath --utts 64 --no-cpu-baseline --steps 1 --warmup 1 --no-pipeline"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1
  python $REPO/tools/rocpd_summary.py $OUT "beam_" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
cp gpurun_out/pmc_${R}_beam_mp/summary.json $O/beam_exact_mp_64_pmc_summary.json 2>/dev/null
ls $O | grep mp
