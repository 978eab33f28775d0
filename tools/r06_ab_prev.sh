#!/bin/bash
# round 6: the product build against build/variants/prev.so (the library before the change at hand) on the four batch lines, twice
set -u
for rep in 1 2; do for v in product prev; do
  if [ $v = product ]; then unset JAMD_LIB; else export JAMD_LIB=build/variants/$v.so; fi
  for w in "e2e512 --workload e2e --utts 512 --steps 4" "dnn256 --workload e2e-dnn --utts 256 --steps 2" "mp512 --workload e2e --multipath --utts 512 --steps 2" "dnnmp256 --workload e2e-dnn --multipath --utts 256 --steps 1"; do
    set -- $w; tag=$1; shift
    timeout 600 python bench.py "$@" --warmup 1 --no-cpu-baseline --no-batch > /dev/null 2>&1
    python -c "
import json; d=json.load(open('bench_detail.json')); print('$v', '$tag', round(d['ms_per_step'],1), round(d['roofline']['beam_kernel_ms'],1), d['pass1']['ok'])"
  done
done; done
