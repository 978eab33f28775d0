#!/bin/bash
# development: sub-step clocks of the exact-order kernel on one C3 utterance, from the probe builds of tools/build_variant.sh
set -u
mkdir -p gpurun_out/xq
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
for v in "$@"; do
  cp build/variants/$v.so julius_amd/libjulius_amd.so
  for u in 1; do
    JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts $u --steps 3 --warmup 1 --no-cpu-baseline --order exact 2>&1 | tail -1 > gpurun_out/xq/probe_${v}_${u}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/xq/probe_${v}_${u}.json"))
print("$v", "utts", $u, "beam_ms", round(d["roofline"]["beam_kernel_ms"],1), "phase_us", d["pass1"]["phase_us_utt0"])
PY
  done
done
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
