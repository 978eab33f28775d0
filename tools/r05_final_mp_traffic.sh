#!/bin/bash
# After the multipath frame's node->token table change: FETCH_SIZE / WRITE_SIZE passes of the three multipath launch shapes
# again, profiles/traffic_first_pass.json regenerated on the box, then the default command under rocprofv3 --kernel-trace.
set -u
REPO=$(pwd); R=r05; O=$REPO/gpurun_out/$R; mkdir -p $O
for f in profiles/r05_*_traffic_pmc_summary.json; do b=${f#profiles/r05_}; cp $f $O/$b; done
NB="--no-cpu-baseline --no-batch"
pmc2() { name=$1; sub=$2; shift 2
  ( OUT=$REPO/gpurun_out/pmc_${R}_$name; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $REPO/bench.py "$@" > $OUT/fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $REPO/bench.py "$@" > $OUT/write.log 2>&1
    python $REPO/tools/rocpd_summary.py $OUT "$sub" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
  cp gpurun_out/pmc_${R}_$name/summary.json $O/${name}_traffic_pmc_summary.json 2>/dev/null; }
pmc2 e2e_mp_512 beam_ --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_mp_256 beam_ --workload e2e --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_dnn_mp_256 beam_ --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
python tools/make_traffic.py r05 > $O/make_traffic.log 2>&1; cat $O/make_traffic.log
cp profiles/traffic_first_pass.json $O/traffic_first_pass.json
bash tools/prof_run.sh r05q_default --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
cp gpurun_out/prof_r05q_default/summary.json $O/default_command_kernel_trace_summary.json; cp gpurun_out/prof_r05q_default/bench_line.json $O/default_command_bench_line_under_rocprof.json
cp bench_detail.json $O/default_command_bench_detail_under_rocprof.json
