#!/bin/bash
set -u
O=gpurun_out/r03x; mkdir -p $O
JAMD_TEST_SHAPE=half timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests_half_shape_everywhere.txt 2>&1; echo "rc=$?" >> $O/gpu_tests_half_shape_everywhere.txt; tail -8 $O/gpu_tests_half_shape_everywhere.txt
