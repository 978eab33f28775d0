#!/bin/bash
set -u
O=gpurun_out/r03y; mkdir -p $O
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
for v in product g8 g16; do
  if [ $v = product ]; then L=""; else L="--lib build/variants/$v.so"; fi
  python tools/xbeam_lab.py run /tmp/xlab $L --tag $v --what c3,c3b,c3c 2>$O/err_$v.txt | tail -1 | tee -a $O/lab.json
  python tools/xbeam_lab.py run /tmp/xlab $L --tag ${v}_half --shape half --what c3 2>>$O/err_$v.txt | tail -1 | tee -a $O/lab.json
done
