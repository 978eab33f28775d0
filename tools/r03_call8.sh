#!/bin/bash
set -u
O=gpurun_out/r03ad; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "rc=$?" >> $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
