#!/bin/bash
# Round-6 baseline on one box: GPU tests, sweep / arrange micro-benchmarks on real C4 frames, the two DNN first-pass lines.
set -u
R=${1:-r06_base}
O=gpurun_out/$R; mkdir -p $O
NB="--no-cpu-baseline --no-batch"
JAMD_SWEEP_PROF=1 timeout 200 python tools/sweep_timing.py > $O/sweep_timing_real_frames.json 2> $O/sweep_timing_phases.txt
JAMD_SWEEP_PROF=1 timeout 200 python tools/arrange_timing.py > $O/arrange_timing_real_frames.json 2> $O/arrange_timing_phases.txt
timeout 600 python bench.py --workload e2e-dnn --utts 256 --steps 2 --warmup 1 $NB > $O/e2e_dnn.log 2>&1; cp bench_detail.json $O/e2e_dnn_detail.json
timeout 600 python bench.py --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 $NB > $O/e2e_dnn_mp.log 2>&1; cp bench_detail.json $O/e2e_dnn_mp_detail.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 $NB > /dev/null 2>&1; cp bench_detail.json $O/e2e_dnn_1_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --multipath --utts 1 --steps 1 --warmup 1 $NB > /dev/null 2>&1; cp bench_detail.json $O/e2e_dnn_mp_1_phases.json
ls $O
