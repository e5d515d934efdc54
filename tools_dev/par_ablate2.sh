#!/bin/bash
# dev (EXCEL_DEV library as tools_dev/ab/dev.so): timing-only arms of the PAR step around the round-3 review's "12-pixel apron" proposal
#   bit3 (8)  = stage 48 of the 64 tile rows (the benefit side: fewer bytes into LDS), bit4 (16) = 8 extra edge-clamped global float2
#   loads per thread per plane (the cost side: the d = 24 taps read from global / L2 instead of LDS); 24 = both.  Results are wrong.
OUT=${1:-gpurun_out/par_arms.txt}
for rep in 1 2; do
for d in 0 8 16 24; do
  EXCEL_PAR_DBG=$d EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 200 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('par dbg=$d', 'par_iterate %.4f' % k['par_iterate'], 'step', d['ms_per_step'])" | tee -a $OUT
done
done
