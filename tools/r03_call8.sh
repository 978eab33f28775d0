#!/bin/bash
set -u
O=gpurun_out/r03s; mkdir -p $O
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
for v in product uprobe product uprobe; do
  if [ $v = product ]; then L=""; else L="--lib build/variants/$v.so"; fi
  python tools/xbeam_lab.py run /tmp/xlab $L --tag $v --what c3,c3b,c3c 2>$O/err_$v.txt | tail -1 | tee -a $O/lab.json
done
JAMD_LIB=build/variants/uprobe.so timeout 600 python -m pytest tests/test_beam_gpu.py tests/test_half_shape_gpu.py tests/test_exact_fuzz_gpu.py -q -m gpu --maxfail=5 > $O/pytest_uprobe.txt 2>&1; echo "rc=$?" >> $O/pytest_uprobe.txt; tail -3 $O/pytest_uprobe.txt
timeout 300 python -m pytest tests/test_loaders_gpu.py -q -m gpu > $O/pytest_loaders.txt 2>&1; tail -2 $O/pytest_loaders.txt
