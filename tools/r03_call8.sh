#!/bin/bash
set -u
O=gpurun_out/r03ai; mkdir -p $O
timeout 900 python -m pytest tests/test_shim_gpu.py tests/test_programs_gpu.py tests/test_loaders_gpu.py -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
