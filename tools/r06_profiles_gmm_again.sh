#!/bin/bash
# re-run of the two gmm passes of tools/r06_profiles.sh with --no-small-T, then the traffic files and the default command again
set -u
R=r06; REPO=$(pwd); O=$REPO/gpurun_out/$R; mkdir -p $O
prof() { name=$1; shift; bash tools/prof_run.sh ${R}_$name "$@" > /dev/null 2>&1;
         cp gpurun_out/prof_${R}_$name/summary.json $O/${name}_kernel_trace_summary.json 2>/dev/null; cp gpurun_out/prof_${R}_$name/bench_line.json $O/${name}_bench_line_under_rocprof.json 2>/dev/null; }
pmc2() { name=$1; sub=$2; shift 2
  ( OUT=$REPO/gpurun_out/pmc_${R}_$name; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $REPO/bench.py "$@" > $OUT/fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $REPO/bench.py "$@" > $OUT/write.log 2>&1
    python $REPO/tools/rocpd_summary.py $OUT "$sub" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
  cp gpurun_out/pmc_${R}_$name/summary.json $O/${name}_traffic_pmc_summary.json 2>/dev/null; }
prof gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline --no-small-T
pmc2 gmm gmm --workload gmm --steps 3 --warmup 1 --no-cpu-baseline --no-small-T
for f in profiles/r06_e2e_*_traffic_pmc_summary.json; do b=$(basename $f); cp $f $O/${b#r06_}; done
python tools/make_traffic.py $R > $O/make_traffic.log 2>&1; cat $O/make_traffic.log
cp profiles/traffic_first_pass.json $O/traffic_first_pass.json; cp profiles/traffic_gmm_tile.json $O/traffic_gmm_tile.json
bash tools/prof_run.sh ${R}_default --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
cp gpurun_out/prof_${R}_default/summary.json $O/default_command_kernel_trace_summary.json; cp gpurun_out/prof_${R}_default/bench_line.json $O/default_command_bench_line_under_rocprof.json
python bench.py > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out > $O/bench_default_final_line.json; cp bench_detail.json $O/bench_default_final_detail.json
wc -c $O/bench_default_final_line.json
