#!/bin/bash
# development: wave 0's clocks inside a level pass of the sweep replay (variant built with -DJAMD_SWEEP_LEVEL_TICKS)
O=gpurun_out/${1:-r06_lv}; mkdir -p $O
JAMD_LIB=build/variants/${2:-lvticks}.so JAMD_SWEEP_PROF=1 timeout 200 python tools/sweep_timing.py > $O/sweep.json 2> $O/phases.txt
cat $O/phases.txt
