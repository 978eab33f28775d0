#!/bin/bash
set -u
O=gpurun_out/r03w; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_half_shape_gpu.py tests/test_prune_order.py tests/test_exact_fuzz_gpu.py tests/test_wide_beam_gpu.py -q -m gpu -x > $O/pytest_$i.txt 2>&1; echo "rc=$?" >> $O/pytest_$i.txt; tail -2 $O/pytest_$i.txt
done
