#!/bin/bash
# same-box A/B of the sweep replay: variant libraries (build/variants/*.so) against the product build, real C4 frames
set -u
R=${1:-r06_ab}; shift
O=gpurun_out/$R; mkdir -p $O
timeout 600 python -m pytest tests/test_prune_order.py tests/test_prune_sweep_gpu.py tests/test_heap_closed_form.py -x -q 2>&1 | tail -4 > $O/prune_tests.txt
for v in product "$@"; do
  if [ $v = product ]; then unset JAMD_LIB; else export JAMD_LIB=build/variants/$v.so; fi
  JAMD_SWEEP_PROF=1 timeout 200 python tools/sweep_timing.py > $O/sweep_$v.json 2> $O/sweep_${v}_phases.txt
  JAMD_SWEEP_PROF=1 timeout 200 python tools/arrange_timing.py > $O/arrange_$v.json 2> $O/arrange_${v}_phases.txt
done
unset JAMD_LIB
cat $O/prune_tests.txt
