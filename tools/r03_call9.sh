#!/bin/bash
set -u
O=gpurun_out/r03h; mkdir -p $O
timeout 600 python -m pytest tests/test_dnn_gpu.py tests/test_beam_gpu.py -q -m gpu -k "dnn or Dnn" --maxfail=10 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout 300 python bench.py --workload dnn --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/dnn.json
python -c "
import json; j=json.load(open('$O/dnn.json')); print('dnn ms', round(j['ms_per_step'],3), 'TF', round(j['roofline']['achieved'],1), 'frac', round(j['roofline']['frac'],4))"
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag product --what c3,c3b,c3c 2>/dev/null | tail -1 | tee $O/lab_product.json
