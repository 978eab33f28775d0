#!/bin/bash
# Round-3 GPU call 3: A7 history kernel, pipelined extraction + give-up rule, C4 end to end timing.
set -u
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_gmm_gpu.py tests/test_prune_order.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py -q -m gpu --maxfail=15 > $O/pytest_a.txt 2>&1; echo "a rc=$?" >> $O/pytest_a.txt; tail -6 $O/pytest_a.txt
timeout 900 python -m pytest tests/test_shim_gpu.py tests/test_beam_gpu.py tests/test_programs_gpu.py tests/test_loaders_gpu.py -q -m gpu --maxfail=15 > $O/pytest_b.txt 2>&1; echo "b rc=$?" >> $O/pytest_b.txt; tail -6 $O/pytest_b.txt
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag product 2>/dev/null | tail -1 | tee $O/lab_product.json
timeout 600 python bench.py --workload e2e-dnn --steps 2 --warmup 1 2> $O/bench_e2e_dnn.err | tail -1 > $O/bench_e2e_dnn.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_1_phases.json
python - <<'PY'
import json
for f in ("bench_e2e_dnn","bench_e2e_dnn_1_phases"):
    try:
        j=json.load(open(f"gpurun_out/r03c/{f}.json"))
        print(f, "ms/step", round(j["ms_per_step"],1), "rtf_inv", round(j["rtf_inv"]), "beam_ms", round(j["roofline"]["beam_kernel_ms"],1), "score_ms", round(j["roofline"]["score_kernels_ms"],1), j.get("parity",{}).get("device_vs_compiled_reference"), j["pass1"])
    except Exception as e: print(f, "ERR", e)
PY
