#!/bin/bash
# same-box A/B: the round-3 tree (build/r03tree, built from commit 8f0aa5b) against the working tree
for rep in 1 2; do
for tree in build/r03tree .; do
  ( cd $tree; a=$(JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), j['pass1']['phase_us_utt0'])")
    b=$(timeout 300 python bench.py --workload e2e --utts 256 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), round(j['roofline']['beam_kernel_ms'],1))")
    echo "$tree | 1 utt: $a | 256 utts: $b" )
done; done
