#!/bin/bash
set -u
O=gpurun_out/${1:-r04h}; mkdir -p $O
JAMD_BEAM_TIMING=1 timeout 600 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>$O/p1.err | tail -1 > $O/bench_e2e_dnn_1_exact_phases.json
JAMD_BEAM_TIMING=1 timeout 600 python bench.py --workload e2e-dnn --flat --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>$O/p2.err | tail -1 > $O/bench_e2e_dnn_flat_1_exact_phases.json
python - <<PY
import json
for f in ("bench_e2e_dnn_1_exact_phases","bench_e2e_dnn_flat_1_exact_phases"):
    try:
        j=json.load(open("$O/"+f+".json"))
        print(f, round(j["ms_per_step"],1), j["pass1"])
    except Exception as e:
        print(f, "failed", e)
PY
