#!/bin/bash
set -u
O=gpurun_out/r03i; mkdir -p $O
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
for v in product orig product orig; do
  if [ $v = product ]; then L=""; else L="--lib build/variants/$v.so"; fi
  python tools/xbeam_lab.py run /tmp/xlab $L --tag $v --what c3,c3b,wide 2>/dev/null | tail -1 | tee -a $O/lab.json
done
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py tests/test_prune_order.py -q -m gpu --maxfail=10 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
