#!/bin/bash
# quick loop for the exact-order kernel: pruning-step fuzz, first-pass parity tests, phase clocks on C3
set -u
mkdir -p gpurun_out/xq
timeout 300 python -m pytest tests/test_prune_order.py -x -q 2>&1 | tail -5 | tee gpurun_out/xq/prune.txt
timeout 600 python -m pytest tests/test_beam_gpu.py -x -q -k "strict_order or edge or streaming or grammar or wordlist" 2>&1 | tail -5 | tee gpurun_out/xq/beam.txt
for u in 1 256; do
  JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts $u --steps 3 --warmup 1 --no-cpu-baseline --order exact 2>&1 | tail -1 > gpurun_out/xq/e2e_${u}_exact.json
  python - <<PY
import json
d=json.load(open("gpurun_out/xq/e2e_${u}_exact.json"))
print("utts", $u, "beam_ms", d["roofline"]["beam_kernel_ms"], "us/frame", d["roofline"]["beam_us_per_frame_per_utt"], "frames/s", d["roofline"]["beam_frames_per_s"], "phase_us", d["pass1"]["phase_us_utt0"])
PY
done
