#!/bin/bash
set -u
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_half_shape_gpu.py tests/test_beam_gpu.py tests/test_wide_beam_gpu.py -q -m gpu --maxfail=5 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag auto --what c3,c3b,c3c,c3d 2>$O/err_auto.txt | tail -1 | tee -a $O/lab.json
python tools/xbeam_lab.py run /tmp/xlab --tag half --shape half --what c3,c3b 2>$O/err_half.txt | tail -1 | tee -a $O/lab.json
