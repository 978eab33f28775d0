#!/bin/bash
# development: A/B timing of library variants (tools/build_variant.sh) on the same box, interleaved, C3 first pass
# usage: tools/exact_ab.sh "1 256" varA varB ...
set -u
utts=$1; shift
mkdir -p gpurun_out/ab
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp build/variants/$v.so julius_amd/libjulius_amd.so
    for u in $utts; do
      timeout 300 python bench.py --workload e2e --utts $u --steps 4 --warmup 1 --no-cpu-baseline --order exact 2>/dev/null | tail -1 > gpurun_out/ab/${v}_${u}_${rep}.json
      python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab/${v}_${u}_${rep}.json")); print("$v", "utts", $u, "rep", $rep, "beam_ms", round(d["roofline"]["beam_kernel_ms"],2), "ok", d["pass1"]["ok"])
except Exception as e: print("$v", $u, "FAILED", e)
PY
    done
  done
done
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
