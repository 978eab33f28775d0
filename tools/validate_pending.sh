#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/validate_pending.sh'): everything that was written after the
# last device time of a round and is therefore opt-in.  Output under gpurun_out/pending/.
# Exits non-zero when any step fails.
set -u -o pipefail
mkdir -p gpurun_out/pending
export JAMD_RUN_UNVALIDATED=1
export JAMD_EXPERIMENTAL_MULTIPATH=1
rc=0
# 1. the multipath strict-order kernel (beam_strict_mp_kernel) and its shim path
timeout 300 python -m pytest tests/test_beam_gpu.py tests/test_shim_gpu.py -k multipath -q 2>&1 | tail -25 | tee gpurun_out/pending/multipath_tests.txt || rc=1
# 2. the whole GPU suite with the opt-in tests included
timeout 900 python -m pytest tests -m gpu -q -rs 2>&1 | tail -40 | tee gpurun_out/pending/gpu_tests_all.txt || rc=1
# 3. cost of the selection stage (K7) and, for comparison, the scoring it follows
PYTHONPATH=. timeout 60 python tools/gms_timing.py 2>&1 | tail -2 | tee gpurun_out/pending/gms_timing.json || rc=1
JAMD_GMS_VARIANT=1 PYTHONPATH=. timeout 60 python tools/gms_timing.py 2>&1 | tail -2 | tee gpurun_out/pending/gms_timing_variant1.json || rc=1
echo "validate_pending rc=$rc" | tee gpurun_out/pending/rc.txt
exit $rc
