#!/bin/bash
# Round-6 evidence run on the GPU box (one call): rocprofv3 kernel traces next to the event-timed bench line of the same
# process, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, MI355X_MICROARCH.md "HBM") of K1 and of the
# first-pass kernels in the launch shapes the default bench runs, SQ counters of the half-shape first pass, and the phase
# clocks of one utterance.  Everything lands under gpurun_out/r06/ (copied to profiles/ by hand afterwards).
set -u
R=${1:-r06}
REPO=$(pwd)
O=$REPO/gpurun_out/$R; mkdir -p $O
NB="--no-cpu-baseline --no-batch"
prof() { name=$1; shift; bash tools/prof_run.sh ${R}_$name "$@" > /dev/null 2>&1;
         cp gpurun_out/prof_${R}_$name/summary.json $O/${name}_kernel_trace_summary.json 2>/dev/null; cp gpurun_out/prof_${R}_$name/bench_line.json $O/${name}_bench_line_under_rocprof.json 2>/dev/null; }
prof gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline --no-small-T
prof dnn --workload dnn --steps 5 --warmup 1 --no-cpu-baseline
prof e2e --workload e2e --utts 512 --steps 2 --warmup 1 $NB
prof e2e_dnn --workload e2e-dnn --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
prof e2e_mp --workload e2e --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
# HBM traffic: one counter per run
pmc2() { name=$1; sub=$2; shift 2
  ( OUT=$REPO/gpurun_out/pmc_${R}_$name; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $REPO/bench.py "$@" > $OUT/fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $REPO/bench.py "$@" > $OUT/write.log 2>&1
    python $REPO/tools/rocpd_summary.py $OUT "$sub" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
  cp gpurun_out/pmc_${R}_$name/summary.json $O/${name}_traffic_pmc_summary.json 2>/dev/null; }
pmc2 gmm gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline --no-small-T
pmc2 e2e_512 beam_ --workload e2e --utts 512 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_256 beam_ --workload e2e --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_dnn_256 beam_ --workload e2e-dnn --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_mp_256 beam_ --workload e2e --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_dnn_mp_256 beam_ --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
# SQ counters of the half-shape first pass (the same two passes as rounds 3-4)
bash tools/prof_beam_pmc.sh ${R}_beam_exact_half 512 > /dev/null 2>&1
cp gpurun_out/pmc_${R}_beam_exact_half/summary.json $O/beam_exact_half_512_pmc_summary.json 2>/dev/null
# phase clocks of one utterance
ph() { name=$1; shift; JAMD_BEAM_TIMING=1 timeout 300 python bench.py "$@" --utts 1 --warmup 1 $NB > /dev/null 2>&1; cp bench_detail.json $O/${name}_1_phases.json; }
ph bench_e2e --workload e2e --steps 3
ph bench_e2e_dnn --workload e2e-dnn --steps 2
ph multipath --workload e2e --multipath --steps 2
ph multipath_dnn --workload e2e-dnn --multipath --steps 1
ls $O
# the six first-pass traffic entries + K1's -> profiles/traffic_*.json (regenerated on the box, copied to profiles/ by hand)
pmc2 e2e_mp_512 beam_ --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline
python tools/make_traffic.py $R > $O/make_traffic.log 2>&1; cat $O/make_traffic.log
cp profiles/traffic_first_pass.json $O/traffic_first_pass.json; cp profiles/traffic_gmm_tile.json $O/traffic_gmm_tile.json
# sweep micro-benchmark on the real C4 frames, GPU suite, the default command under the kernel trace and plain
JAMD_SWEEP_PROF=1 timeout 200 python tools/sweep_timing.py > $O/sweep_timing_real_frames.json 2> $O/sweep_timing_phases.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gpu_tests_final.txt
bash tools/prof_run.sh ${R}_default --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
cp gpurun_out/prof_${R}_default/summary.json $O/default_command_kernel_trace_summary.json; cp gpurun_out/prof_${R}_default/bench_line.json $O/default_command_bench_line_under_rocprof.json
python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out > $O/bench_default_final_line.json; cp bench_detail.json $O/bench_default_final_detail.json
cat $O/gpu_tests_final.txt; wc -c $O/bench_default_final_line.json
