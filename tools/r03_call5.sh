#!/bin/bash
# Round-3 GPU call 5: dnn_lse frames-per-wave sweep, DNN bit-exactness, jamd_batch pipelining, default bench line.
set -u
O=gpurun_out/r03e; mkdir -p $O
timeout 600 python -m pytest tests/test_dnn_gpu.py tests/test_loaders_gpu.py tests/test_wide_beam_gpu.py -q -m gpu --maxfail=10 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for l in 4 8 16 32 64; do
  JAMD_LIB=build/variants/dnndev.so JAMD_LSE_LPW=$l timeout 300 python bench.py --workload dnn --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/dnn_lpw$l.json
  python -c "
import json; j=json.load(open('$O/dnn_lpw$l.json')); print('lpw', $l, 'ms', round(j['ms_per_step'],3), 'TF', round(j['roofline']['achieved'],1), 'frac', round(j['roofline']['frac'],4))"
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03e/bench_default.json"))
print("C2", j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j.get("parity_spot_check"))
for k in ("e2e","e2e_strong","e2e_dnn","dnn"):
    v=j[k]; print(k, "ms/step", round(v["ms_per_step"],1), "rtf_inv", round(v["rtf_inv"]), v["roofline"].get("beam_kernel_ms"), v["roofline"].get("frac"), v.get("parity",{}).get("device_vs_compiled_reference",{}).get("trellis_identical"), v.get("parity_spot_check"))
PY
