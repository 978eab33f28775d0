#!/bin/bash
# sweep replay: first run on the GPU
set -u
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_prune_sweep_gpu.py -x -q > $O/sweep_tests.txt 2>&1; tail -15 $O/sweep_tests.txt
timeout 900 python -m pytest tests/test_prune_order.py tests/test_wide_beam_gpu.py -x -q > $O/prune_tests.txt 2>&1; tail -5 $O/prune_tests.txt
