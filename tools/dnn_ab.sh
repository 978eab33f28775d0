#!/bin/bash
# development: A/B timing of library variants on the DNN scoring workload (same box, interleaved) + parity of the last one
set -u
cp julius_amd/libjulius_amd.so /tmp/lib_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp build/variants/$v.so julius_amd/libjulius_amd.so
    python bench.py --workload dnn --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'rep', $rep, 'ms', round(d['ms_per_step'],3), 'TFLOPs', round(d['roofline']['achieved'],2), 'parity', d.get('parity_spot_check', d.get('parity'))) "
  done
done
cp /tmp/lib_keep.so julius_amd/libjulius_amd.so
