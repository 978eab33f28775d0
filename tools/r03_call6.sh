#!/bin/bash
# Round-3 GPU call 6: whole GPU suite on the current tree + strong line (longest-first launch order) + lab.
set -u
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --maxfail=20 > $O/pytest_gpu.txt 2>&1; echo "suite rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
timeout 300 python bench.py --workload e2e --strong --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_strong.json
timeout 300 python bench.py --workload e2e --utts 1024 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_1024.json
python - <<'PY'
import json
for f in ("bench_e2e_strong","bench_e2e_1024"):
    try:
        j=json.load(open(f"gpurun_out/r03f/{f}.json"))
        print(f, "ms/step", round(j["ms_per_step"],1), "rtf_inv", round(j["rtf_inv"]), "beam_ms", round(j["roofline"]["beam_kernel_ms"],1), "score_ms", round(j["roofline"]["score_kernels_ms"],1), j["pass1"]["ok"])
    except Exception as e: print(f, "ERR", e)
PY
