#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py tests/test_shim_gpu.py -q -m gpu --maxfail=10 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag product --what c3,c3b,c3c,wide 2>/dev/null | tail -1 | tee $O/lab_product.json
