#!/bin/bash
set -u
O=gpurun_out/r03ah; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03ah/bench_default.json"))
print("C2", round(j["ms_per_step"],1), j["roofline"]["frac"])
for k in ("e2e","e2e_strong","e2e_256","e2e_dnn","dnn"):
    v=j[k]; print(k, "ms/step", round(v["ms_per_step"],1), "rtf_inv", round(v["rtf_inv"]), 'score', v["roofline"].get("score_kernels_ms"), 'beam', v["roofline"].get("beam_kernel_ms"), v["roofline"].get("frac"), v.get("parity",{}).get("device_vs_compiled_reference",{}).get("trellis_identical"))
PY
