#!/bin/bash
# round 5 experiment: K1 (gmm_tile) co-resident with the half-shape first pass.  Variant xe_half5 = half shape compiled for 5
# waves per SIMD (96 VGPRs); JAMD_HALF_LDS_KB=62 leaves the LDS of one gmm_tile workgroup per CU.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/coreside
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
run() { tag=$1; shift; timeout 300 python bench.py --workload e2e --utts 512 --steps 6 --warmup 1 --no-cpu-baseline --no-batch "$@" > /dev/null 2>&1; cp bench_detail.json gpurun_out/coreside/$tag.json
  python - $tag <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/coreside/{sys.argv[1]}.json"))
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],2), "beam_ms", round(d["roofline"]["beam_kernel_ms"],2), "score_ms", round(d["roofline"]["score_kernels_ms"],2), "ok", d["pass1"]["ok"], d["roofline"]["beam_kernel_ms_steps"])
PY
}
run main_pipelined
run main_one_stream --no-pipeline
cp build/variants/xe_half5.so julius_amd/libjulius_amd.so
JAMD_HALF_LDS_KB=79 run half5_lds79
JAMD_HALF_LDS_KB=62 run half5_lds62
JAMD_HALF_LDS_KB=62 JAMD_RESIDENT_SHARE=50 run half5_lds62_share50
JAMD_HALF_LDS_KB=62 run half5_lds62_one_stream --no-pipeline
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
