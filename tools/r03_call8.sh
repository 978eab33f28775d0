#!/bin/bash
set -u
O=gpurun_out/r03ag; mkdir -p $O
for n in 256 64; do for m in 1 100000; do
  JAMD_BENCH_TF_MIN=$m timeout 600 python bench.py --workload e2e --utts $n --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/u${n}_m$m.json
  python - <<PY
import json
r=json.load(open("gpurun_out/r03ag/u${n}_m$m.json")); print("utts $n tf_min $m", round(r['ms_per_step'],1), 'score', round(r['roofline']['score_kernels_ms'],1), 'beam', round(r['roofline']['beam_kernel_ms'],1))
PY
done; done
