#!/bin/bash
# DNN scoring kernels: parity tests, then kernel times under rocprofv3 (64 000 frames)
set -u
mkdir -p gpurun_out/dq
timeout 300 python -m pytest tests/test_dnn_gpu.py tests/test_loaders_gpu.py -q -k "dnn" 2>&1 | tail -3
python bench.py --workload dnn --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'TFLOPs', d['roofline']['achieved'], 'frac', d['roofline']['frac'])"
bash tools/prof_run.sh dnnq --workload dnn --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/prof_dnnq/summary.json"))
for k in d.get("kernels", d if isinstance(d,list) else [])[:8] if not isinstance(d,dict) or "kernels" in d else []:
    print(k)
print(json.dumps(d)[:1500])
PY
