#!/bin/bash
# round 6: memory-side traffic of the first-pass launch shapes again (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs),
# profiles/traffic_first_pass.json regenerated on the box (K1's entry is carried over from round 5: the kernel is unchanged).
set -u
REPO=$(pwd); R=r06; O=$REPO/gpurun_out/$R; mkdir -p $O
cp profiles/r05_gmm_traffic_pmc_summary.json $O/gmm_traffic_pmc_summary.json
NB="--no-cpu-baseline --no-batch"
pmc2() { name=$1; sub=$2; shift 2
  ( OUT=$REPO/gpurun_out/pmc_${R}_$name; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc -- python $REPO/bench.py "$@" > $OUT/fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc -- python $REPO/bench.py "$@" > $OUT/write.log 2>&1
    python $REPO/tools/rocpd_summary.py $OUT "$sub" > $OUT/summary.json 2>/dev/null; find $OUT -name "*.db" -delete )
  cp gpurun_out/pmc_${R}_$name/summary.json $O/${name}_traffic_pmc_summary.json 2>/dev/null; }
pmc2 e2e_512 beam_ --workload e2e --utts 512 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_256 beam_ --workload e2e --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_dnn_256 beam_ --workload e2e-dnn --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_mp_512 beam_ --workload e2e --multipath --utts 512 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_mp_256 beam_ --workload e2e --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
pmc2 e2e_dnn_mp_256 beam_ --workload e2e-dnn --multipath --utts 256 --steps 1 --warmup 1 $NB --no-pipeline
python tools/make_traffic.py r06 > $O/make_traffic.log 2>&1; cat $O/make_traffic.log
cp profiles/traffic_first_pass.json $O/traffic_first_pass.json
