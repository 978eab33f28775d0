#!/bin/bash
# Round-3 evidence run on the GPU box: full GPU suite, the default bench line, rocprofv3 kernel traces (each next to the
# event-timed line of the SAME process and to the un-profiled line of the same command), PMC passes for the first pass.
set -u
R=${1:-r03}
O=gpurun_out/$R; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
prof() { name=$1; shift; timeout 600 python bench.py "$@" 2>/dev/null | tail -1 > $O/${name}_bench_line_unprofiled.json; bash tools/prof_run.sh ${R}_$name "$@" > /dev/null 2>&1;
         cp gpurun_out/prof_${R}_$name/summary.json $O/${name}_kernel_trace_summary.json 2>/dev/null; cp gpurun_out/prof_${R}_$name/bench_line.json $O/${name}_bench_line_under_rocprof.json 2>/dev/null; }
prof gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline
prof e2e --workload e2e --utts 512 --steps 2 --warmup 1 --no-cpu-baseline
prof e2e_256 --workload e2e --utts 256 --steps 2 --warmup 1 --no-cpu-baseline
prof e2e_dnn --workload e2e-dnn --utts 256 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline
prof dnn --workload dnn --steps 5 --warmup 1 --no-cpu-baseline
bash tools/prof_beam_pmc.sh ${R}_beam_exact_half 512 > /dev/null 2>&1
cp gpurun_out/pmc_${R}_beam_exact_half/summary.json $O/beam_exact_half_512_pmc_summary.json 2>/dev/null
bash tools/prof_beam_pmc.sh ${R}_beam_exact 256 > /dev/null 2>&1
cp gpurun_out/pmc_${R}_beam_exact/summary.json $O/beam_exact_pmc_summary.json 2>/dev/null
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_1_exact_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_1_exact_phases.json
timeout 300 python bench.py --workload e2e --utts 64 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_64_per_gpu.json
ls $O
python - <<PY
import json
j=json.load(open("$O/bench_default.json"))
print("C2", round(j["ms_per_step"],1), j["value"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j.get("parity_spot_check"))
for k in ("e2e","e2e_strong","e2e_256","e2e_dnn","dnn"):
    v=j[k]; print(k, "ms/step", round(v["ms_per_step"],1), "rtf_inv", round(v["rtf_inv"]), v["roofline"].get("beam_kernel_ms"), v["roofline"].get("frac"), v.get("parity",{}).get("device_vs_compiled_reference",{}).get("trellis_identical"), v.get("parity_spot_check"))
PY
