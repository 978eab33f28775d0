#!/bin/bash
# development A/B on one box: the multipath frame's per-node token table over all nodes (build/variants/mp_nodetok_old.so)
# against the table over the nodes a root leads to (the shipped library)
set -u
O=gpurun_out/r05n; mkdir -p $O
python -m pytest tests/test_multipath_exact_gpu.py tests/test_half_shape_gpu.py tests/test_forward_dfa_gpu.py tests/test_full_size_parity_gpu.py -x -q > $O/mp_tests.txt 2>&1; tail -2 $O/mp_tests.txt
cp julius_amd/libjulius_amd.so /tmp/new.so
for v in new old new2; do
  if [ $v = old ]; then cp build/variants/mp_nodetok_old.so julius_amd/libjulius_amd.so; else cp /tmp/new.so julius_amd/libjulius_amd.so; fi
  python bench.py --workload e2e --multipath --utts 512 --steps 3 --warmup 1 --no-cpu-baseline --no-batch 2>/dev/null | tail -1 > $O/e2e_mp_512_$v.json
  if [ $v != new2 ]; then python bench.py --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 --no-cpu-baseline --no-batch 2>/dev/null | tail -1 > $O/e2e_dnn_mp_256_$v.json; fi
done
cp /tmp/new.so julius_amd/libjulius_amd.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05n/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"], d.get("beam_kernel_ms"), d.get("parity"), d.get("pass1_ok"))
    except Exception as e: print(f, "ERR", e)
PY
