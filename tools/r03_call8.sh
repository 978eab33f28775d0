#!/bin/bash
set -u
O=gpurun_out/r03r; mkdir -p $O
for mode in serial pipe pipe_prio; do
  case $mode in serial) E="";; pipe) E="JAMD_BENCH_FORCE_PIPE=1";; pipe_prio) E="JAMD_BENCH_FORCE_PIPE=1 JAMD_BENCH_PRIO=1";; esac
  env $E timeout 600 python bench.py --workload e2e --utts 512 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$mode.json
  env $E timeout 600 python bench.py --workload e2e --utts 256 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${mode}_256.json
  python - <<PY
import json
for f in ("$mode","${mode}_256"):
    r=json.load(open("gpurun_out/r03r/%s.json"%f)); print(f, round(r['ms_per_step'],1), 'score', round(r['roofline']['score_kernels_ms'],1), 'beam', round(r['roofline']['beam_kernel_ms'],1), r['config']['pipelined'][:8])
PY
done
