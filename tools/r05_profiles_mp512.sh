set -u
REPO=$(pwd); R=r05; O=$REPO/gpurun_out/$R; mkdir -p $O
NB="--no-cpu-baseline --no-batch"
OUT=$REPO/gpurun_out/pmc_${R}_e2e_mp_512; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $REPO/bench.py --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $REPO/bench.py --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline > $OUT/write.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT "beam_" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete
cp $OUT/summary.json $O/e2e_mp_512_traffic_pmc_summary.json
cd $REPO; bash tools/prof_run.sh ${R}_e2e_mp512 --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline > /dev/null 2>&1
cp gpurun_out/prof_${R}_e2e_mp512/summary.json $O/e2e_mp_512_kernel_trace_summary.json; cp gpurun_out/prof_${R}_e2e_mp512/bench_line.json $O/e2e_mp_512_bench_line_under_rocprof.json
python - <<PY
import json
d=json.load(open("$O/e2e_mp_512_traffic_pmc_summary.json"))
for k,v in d.items(): print(k, v.get("pmc_avg_per_dispatch"), [(x["name"][28:70], x["calls"], round(x["avg_us"]/1e3,1)) for x in v.get("kernels",[])[:2]])
PY
