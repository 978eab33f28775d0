#!/bin/bash
# First-pass order modes on the C3 workload (bench.py --workload e2e): exact (default), fast, exact with the
# sequential extraction loop, strict (one lane per utterance).  Output: gpurun_out/modes/*.json
set -u
mkdir -p gpurun_out/modes
for m in exact fast exact_serial; do
  JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 256 --steps 3 --warmup 1 --no-cpu-baseline --order $m 2>&1 | tail -1 > gpurun_out/modes/e2e_256_$m.json
done
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline --order exact 2>&1 | tail -1 > gpurun_out/modes/e2e_1_exact.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline --order fast 2>&1 | tail -1 > gpurun_out/modes/e2e_1_fast.json
timeout 600 python bench.py --workload e2e --utts 64 --steps 1 --warmup 0 --no-cpu-baseline --order strict 2>&1 | tail -1 > gpurun_out/modes/e2e_64_strict.json
tail -c 700 gpurun_out/modes/*.json
