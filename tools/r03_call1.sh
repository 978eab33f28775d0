#!/bin/bash
# Round-3 GPU call 1: wide-beam exact-order kernel (new tests first, under their own timeout), whole GPU suite, default bench line.
set -u
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_wide_beam_gpu.py -q -m gpu --maxfail=30 > gpurun_out/r03a/pytest_wide.txt 2>&1
echo "wide rc=$?" >> gpurun_out/r03a/pytest_wide.txt
tail -25 gpurun_out/r03a/pytest_wide.txt
timeout 900 python -m pytest tests -q -m gpu --maxfail=30 --deselect tests/test_wide_beam_gpu.py > gpurun_out/r03a/pytest_gpu.txt 2>&1
echo "suite rc=$?" >> gpurun_out/r03a/pytest_gpu.txt
tail -25 gpurun_out/r03a/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r03a/bench_default.err
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03a/bench_e2e_dnn_1_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03a/bench_e2e_1_phases.json
python - <<'PY'
import json
for f in ("bench_default","bench_e2e_dnn_1_phases","bench_e2e_1_phases"):
    try:
        j=json.load(open(f"gpurun_out/r03a/{f}.json"))
        print(f, j.get("ms_per_step"), j.get("value"), {k:(v.get("ms_per_step"), v.get("roofline",{}).get("beam_kernel_ms"), v.get("parity",{}).get("device_vs_compiled_reference")) for k,v in j.items() if isinstance(v,dict) and "ms_per_step" in v})
        print("  roofline", j.get("roofline",{}).get("frac"), j.get("roofline",{}).get("kernel_ms"), j.get("pass1"))
    except Exception as e: print(f, "ERR", e)
PY
