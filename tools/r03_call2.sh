#!/bin/bash
# Round-3 GPU call 2: pipelined extraction + i_last cut-off (product), thresholded-heap variant, fine phase probes.
set -u
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
timeout 600 python -m pytest tests/test_prune_order.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py tests/test_heap_closed_form.py -q -m gpu --maxfail=10 > $O/pytest_product.txt 2>&1; echo "product rc=$?" >> $O/pytest_product.txt; tail -4 $O/pytest_product.txt
JAMD_LIB=build/variants/thresh.so timeout 600 python -m pytest tests/test_prune_order.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py -q -m gpu --maxfail=10 > $O/pytest_thresh.txt 2>&1; echo "thresh rc=$?" >> $O/pytest_thresh.txt; tail -4 $O/pytest_thresh.txt
timeout 600 python -m pytest tests/test_beam_gpu.py tests/test_gmm_gpu.py tests/test_dnn_gpu.py -q -m gpu --maxfail=10 > $O/pytest_more.txt 2>&1; echo "more rc=$?" >> $O/pytest_more.txt; tail -4 $O/pytest_more.txt
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag product 2>/dev/null | tail -1 | tee $O/lab_product.json
python tools/xbeam_lab.py run /tmp/xlab --lib build/variants/thresh.so 2>/dev/null | tail -1 | tee $O/lab_thresh.json
for p in 1 2 3 5; do python tools/xbeam_lab.py run /tmp/xlab --lib build/variants/probe$p.so --what c3,wide 2>/dev/null | tail -1 | tee $O/lab_probe$p.json; done
python tools/xbeam_lab.py run /tmp/xlab --tag fast --order fast --what c3,c3b,wide,wideb 2>/dev/null | tail -1 | tee $O/lab_fast.json
