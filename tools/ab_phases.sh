#!/bin/bash
# development: single-utterance phase clocks of library variants (tools/build_variant.sh), interleaved on one box
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp build/variants/$v.so julius_amd/libjulius_amd.so
    a=$(JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1), j['pass1']['phase_us_utt0'])")
    b=$(timeout 300 python bench.py --workload e2e --utts 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],1))")
    echo "$v | timed: $a | untimed: $b"
  done
done
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
