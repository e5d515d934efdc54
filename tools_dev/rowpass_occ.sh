#!/bin/bash
# dev: row-pass time with 3 / 2 / 1 workgroups per CU.  Needs, in excel_launch_attn_rowpass (attn.hip), the launch's dynamic LDS size taken from
# the environment in a dev build:   int dyn = getenv("EXCEL_ROWPASS_LDS") ? atoi(getenv("EXCEL_ROWPASS_LDS")) : 0;  hipLaunchKernelGGL(..., dyn, st, a);
# (the padding only lowers occupancy; round 3 measured 2.73 / 3.07 / 4.56 ms per step)
for l in 0 20000 40000; do
  EXCEL_ROWPASS_LDS=$l timeout 120 python bench.py --cpu-images 0 --ragged-images 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('lds+$l', 'rowpass', k['attn_rowpass'], 'step', d['ms_per_step'])"
done
