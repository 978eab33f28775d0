#!/bin/bash
# Round-2 evidence run on the GPU box: final bench lines + rocprofv3 kernel traces + PMC passes for the exact-order kernel.
set -u
mkdir -p gpurun_out/r02
# 1. bench lines
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
for m in exact fast; do
  python bench.py --workload e2e --utts 256 --steps 4 --warmup 1 --no-cpu-baseline --order $m 2>/dev/null | tail -1 > gpurun_out/r02/bench_e2e_256_$m.json
done
python bench.py --workload e2e --utts 512 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/bench_e2e_512_exact.json
JAMD_BEAM_TIMING=1 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/bench_e2e_1_exact_phases.json
python bench.py --workload e2e-dnn --utts 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/bench_e2e_dnn_256.json
# 2. kernel traces (rocprofv3 --kernel-trace --stats)
bash tools/prof_run.sh r02_gmm --workload gmm --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
bash tools/prof_run.sh r02_e2e --workload e2e --utts 256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
bash tools/prof_run.sh r02_dnn --workload dnn --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
# 3. PMC passes for the exact-order first-pass kernel (own runs, no trace domains)
bash tools/prof_beam_pmc.sh r02_beam_exact 256 > /dev/null 2>&1
ls gpurun_out/r02 gpurun_out/prof_r02_* gpurun_out/pmc_r02_beam_exact
