#!/bin/bash
set -u
O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests/test_loaders_gpu.py tests/test_shim_gpu.py -q -m gpu --maxfail=5 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
