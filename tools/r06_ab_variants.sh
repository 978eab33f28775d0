#!/bin/bash
set -u
O=gpurun_out/$1; shift; mkdir -p $O; summ=$O/summary.txt; : > $summ
run() { local v=$1 tag=$2; shift 2
  export JAMD_LIB=build/variants/$v.so
  timeout 600 python bench.py "$@" --no-cpu-baseline --no-batch > $O/${v}_${tag}.out 2> $O/${v}_${tag}.err
  cp bench_detail.json $O/${v}_${tag}.json 2>/dev/null
  python - "$O" "$v" "$tag" >> $summ <<'PY'
import json, sys
o, v, tag = sys.argv[1:4]
try:
    r = json.load(open(f"{o}/{v}_{tag}.json")); p1 = r.get("pass1", {})
    print(v, tag, "ms_per_step", round(r["ms_per_step"], 2), "beam_ms", round(r["roofline"].get("beam_kernel_ms", 0), 2), "ok", p1.get("ok"), p1.get("phase_us_utt0"))
except Exception as e:
    print(v, tag, "FAILED", repr(e))
PY
}
for rep in 1 2; do for v in "$@"; do
  run $v dnn256_$rep --workload e2e-dnn --utts 256 --steps 2 --warmup 1
  run $v e2e512_$rep --workload e2e --utts 512 --steps 3 --warmup 1
done; done
for v in "$@"; do JAMD_BEAM_TIMING=1 run $v dnn1 --workload e2e-dnn --utts 1 --steps 2 --warmup 1; done
cat $summ
