#!/bin/bash
# round 5: A/B of the exact-order first-pass kernels' register diet on one box (variants from tools/build_variant.sh in
# build/variants/, built on the CPU side), plus K1's log-sum step.  Writes gpurun_out/ab/<variant>_<run>.json (the bench
# detail tree) and a summary table gpurun_out/ab/summary.txt.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
summ=gpurun_out/ab/summary.txt
: > $summ
run() {  # variant tag args...
  local v=$1 tag=$2; shift 2
  timeout 400 python bench.py "$@" --no-cpu-baseline --no-batch > gpurun_out/ab/${v}_${tag}.out 2> gpurun_out/ab/${v}_${tag}.err
  cp bench_detail.json gpurun_out/ab/${v}_${tag}.json 2>/dev/null
  python - "$v" "$tag" >> $summ <<'PY'
import json, sys
v, tag = sys.argv[1:3]
try:
    d = json.load(open(f"gpurun_out/ab/{v}_{tag}.json"))
    k = [x for x in d if isinstance(d[x], dict) and "ms_per_step" in d[x]]
    r = d if "roofline" in d and "beam_kernel_ms" in d["roofline"] else d[k[0]]
    p1 = r.get("pass1", {})
    print(v, tag, "ms_per_step", round(r["ms_per_step"], 2), "beam_ms", round(r["roofline"].get("beam_kernel_ms", 0), 2),
          "score_ms", round(r["roofline"].get("score_kernels_ms", 0), 2), "ok", p1.get("ok"), "phase_us", p1.get("phase_us_utt0"))
except Exception as e:
    print(v, tag, "FAILED", repr(e))
PY
  tail -1 $summ
}
for v in "$@"; do
  cp build/variants/$v.so julius_amd/libjulius_amd.so
  run $v c3_512 --workload e2e --utts 512 --steps 4 --warmup 1
  JAMD_BEAM_TIMING=1 run $v c3_1 --workload e2e --utts 1 --steps 3 --warmup 1
  run $v c3mp_256 --workload e2e --multipath --utts 256 --steps 2 --warmup 1
  run $v c4_256 --workload e2e-dnn --utts 256 --steps 2 --warmup 1
  run $v c4mp_256 --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1
done
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
