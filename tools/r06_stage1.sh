#!/bin/bash
O=gpurun_out/r06_o; mkdir -p $O
timeout 1500 python -m pytest tests/test_prune_order.py tests/test_prune_sweep_gpu.py tests/test_heap_closed_form.py tests/test_wide_beam_gpu.py tests/test_multipath_exact_gpu.py tests/test_half_shape_gpu.py tests/test_exact_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
NB="--no-cpu-baseline --no-batch"
timeout 600 python bench.py --workload e2e-dnn --utts 256 --steps 2 --warmup 1 $NB > $O/e2e_dnn.log 2>&1; cp bench_detail.json $O/e2e_dnn_detail.json
timeout 600 python bench.py --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 $NB > $O/e2e_dnn_mp.log 2>&1; cp bench_detail.json $O/e2e_dnn_mp_detail.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 $NB > /dev/null 2>&1; cp bench_detail.json $O/e2e_dnn_1_phases.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --multipath --utts 1 --steps 1 --warmup 1 $NB > /dev/null 2>&1; cp bench_detail.json $O/e2e_dnn_mp_1_phases.json
cat $O/tests.txt
python - <<'P'
import json
for f in ['e2e_dnn_detail','e2e_dnn_mp_detail','e2e_dnn_1_phases','e2e_dnn_mp_1_phases']:
    d=json.load(open(f'gpurun_out/r06_o/{f}.json'))
    print(f, 'ms_per_step', round(d['ms_per_step'],1), 'rtf_inv', round(d.get('rtf_inv',0),1), 'beam_ms', round(d['roofline']['beam_kernel_ms'],1), d['pass1'].get('phase_us_utt0'), d['pass1'].get('prune_paths_utt0'))
P
