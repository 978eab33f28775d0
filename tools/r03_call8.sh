#!/bin/bash
set -u
O=gpurun_out/r03ab; mkdir -p $O
timeout 600 python -m pytest tests/test_half_shape_gpu.py tests/test_exact_fuzz_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -12 $O/pytest.txt
