#!/bin/bash
# Round-3 GPU call 4: pipelined extraction (branch-free) timing + fuzz, counter passes for K1 (VALU/clock) and K3 (MFMA busy).
set -u
O=gpurun_out/r03d; mkdir -p $O
REPO=$(pwd)
timeout 600 python -m pytest tests/test_prune_order.py tests/test_wide_beam_gpu.py tests/test_exact_fuzz_gpu.py -q -m gpu --maxfail=10 > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
python tools/xbeam_lab.py prepare /tmp/xlab > /dev/null 2>&1
python tools/xbeam_lab.py run /tmp/xlab --tag product 2>/dev/null | tail -1 | tee $O/lab_product.json
JAMD_BEAM_TIMING=1 timeout 300 python bench.py --workload e2e-dnn --utts 1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_1_phases.json
timeout 300 python bench.py --workload e2e-dnn --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_dnn_256.json
timeout 300 python bench.py --workload e2e --strong --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_strong.json
timeout 300 python bench.py --workload e2e --strong --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline 2>/dev/null | tail -1 > $O/bench_e2e_strong_nopipe.json
timeout 300 python bench.py --workload e2e --utts 64 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_e2e_64.json
timeout 300 python bench.py --workload e2e --utts 64 --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline 2>/dev/null | tail -1 > $O/bench_e2e_64_nopipe.json
python - <<'PY'
import json
for f in ("bench_e2e_dnn_1_phases","bench_e2e_dnn_256","bench_e2e_strong","bench_e2e_strong_nopipe","bench_e2e_64","bench_e2e_64_nopipe"):
    try:
        j=json.load(open(f"gpurun_out/r03d/{f}.json"))
        print(f, "ms/step", round(j["ms_per_step"],1), "rtf_inv", round(j["rtf_inv"]), "beam_ms", round(j["roofline"]["beam_kernel_ms"],1), "score_ms", round(j["roofline"]["score_kernels_ms"],1), j["pass1"]["phase_us_utt0"])
    except Exception as e: print(f, "ERR", e)
PY
# counters: own --pmc runs, no trace domains
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; sub=$2; shift 2; args="$1"; shift; mkdir -p $REPO/$O/pmc_$name; rocprofv3 --pmc "$@" -d $REPO/$O/pmc_$name/p -o pmc -- python $REPO/bench.py $args > $REPO/$O/pmc_$name/log.txt 2>&1; python $REPO/tools/rocpd_summary.py $REPO/$O/pmc_$name "$sub" > $REPO/$O/pmc_$name.json 2>/dev/null; find $REPO/$O/pmc_$name -name "*.db" -delete; }
pmc gmm_a gmm_tile "--workload gmm --steps 3 --warmup 1 --no-cpu-baseline" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc gmm_b gmm_tile "--workload gmm --steps 3 --warmup 1 --no-cpu-baseline" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT
pmc dnn_a dnn_ "--workload dnn --steps 3 --warmup 1 --no-cpu-baseline" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES
pmc dnn_b dnn_ "--workload dnn --steps 3 --warmup 1 --no-cpu-baseline" SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03d/pmc_*.json")):
    try:
        j=json.load(open(f))
        for k,v in j.items():
            print(f.split("/")[-1], {kk:{c:round(x) for c,x in vv.items()} for kk,vv in v.get("pmc_avg_per_dispatch",{}).items()}, [ (x["name"][:40], round(x["avg_us"],1)) for x in v.get("kernels",[])[:4]])
    except Exception as e: print(f,"ERR",e)
PY
