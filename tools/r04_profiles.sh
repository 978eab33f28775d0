#!/bin/bash
# Round-4 evidence run on the GPU box: full GPU suite, the default bench line, rocprofv3 kernel traces (each next to the
# event-timed line of the SAME process and to the un-profiled line of the same command), PMC passes (first pass at beam
# 800 in both shapes, first pass at beam 4000 with the sweep replay, the GMM kernel's HBM traffic), phase clocks and
# pruning-path counters of one utterance, the sweep's phase clocks on real frames, the kernel timeline of a pipelined step.
set -u
R=${1:-r04}
O=gpurun_out/$R; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
prof() { name=$1; shift; timeout 600 python bench.py "$@" 2>/dev/null | tail -1 > $O/${name}_bench_line_unprofiled.json; bash tools/prof_run.sh ${R}_$name "$@" > /dev/null 2>&1;
         cp gpurun_out/prof_${R}_$name/summary.json $O/${name}_kernel_trace_summary.json 2>/dev/null; cp gpurun_out/prof_${R}_$name/bench_line.json $O/${name}_bench_line_under_rocprof.json 2>/dev/null; }
prof gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline
prof e2e --workload e2e --utts 512 --steps 2 --warmup 1 --no-cpu-baseline
prof e2e_dnn --workload e2e-dnn --utts 256 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline
prof dnn --workload dnn --steps 5 --warmup 1 --no-cpu-baseline
bash tools/prof_beam_pmc.sh ${R}_beam_exact_half 512 > /dev/null 2>&1
cp gpurun_out/pmc_${R}_beam_exact_half/summary.json $O/beam_exact_half_512_pmc_summary.json 2>/dev/null
# the first pass at beam 4000 (C4 input that decodes): counters of the kernel that holds the sweep replay
( REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_${R}_beam_dnn; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
  BENCH="python $REPO/bench.py --workload e2e-dnn --utts 64 --no-cpu-baseline --steps 1 --warmup 1 --no-pipeline"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/a -o pmc -- $BENCH > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $OUT/b -o pmc -- $BENCH > $OUT/b.log 2>&1
  python $REPO/tools/rocpd_summary.py $OUT "beam_" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
cp gpurun_out/pmc_${R}_beam_dnn/summary.json $O/beam_exact_dnn_64_pmc_summary.json 2>/dev/null
# HBM traffic of the GMM kernel as it is now (separate passes, MI355X_MICROARCH.md "HBM")
( REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_${R}_gmm; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
  BENCH="python $REPO/bench.py --workload gmm --steps 3 --warmup 1 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq -o pmc -- $BENCH > $OUT/sq.log 2>&1
  python $REPO/tools/rocpd_summary.py $OUT "gmm" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
cp gpurun_out/pmc_${R}_gmm/summary.json $O/gmm_pmc_summary.json 2>/dev/null
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_1_exact_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_1_exact_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --flat --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_flat_1_exact_phases.json
timeout 300 python bench.py --workload e2e --utts 64 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_64_per_gpu.json
# the multipath frame (csrc/beam_exact_mp.h): phase clocks of one utterance, C3 task with -multipath and the DNN recipe (-b 4000 -multipath)
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --multipath --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/multipath_1_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --multipath --utts 1 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/multipath_dnn_1_phases.json
JAMD_SWEEP_PROF=1 timeout 120 python tools/arrange_timing.py > $O/arrange_timing_real_frames.json 2> $O/arrange_timing_phases.txt
JAMD_SWEEP_PROF=1 timeout 120 python tools/sweep_timing.py > $O/sweep_timing_real_frames.json 2> $O/sweep_timing_phases.txt
bash tools/kernel_timeline.sh ${R}_e2e512 --workload e2e --utts 512 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp gpurun_out/timeline_${R}_e2e512/timeline.txt $O/e2e_512_kernel_timeline.txt 2>/dev/null
ls $O
python - <<PY
import json
j=json.load(open("$O/bench_default.json"))
print("C2", round(j["ms_per_step"],1), j["value"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j.get("parity_spot_check"))
for k in ("e2e","e2e_strong","e2e_256","e2e_mp","e2e_dnn","e2e_dnn_strong","e2e_dnn_mp","e2e_dnn_flat","dnn"):
    v=j.get(k)
    if not v: print(k, "missing"); continue
    print(k, "ms/step", round(v["ms_per_step"],1), "rtf_inv", round(v["rtf_inv"]), v["roofline"].get("beam_kernel_ms"), v["roofline"].get("frac"), v.get("parity",{}).get("device_vs_compiled_reference",{}).get("trellis_identical"), v.get("parity_spot_check"))
PY
