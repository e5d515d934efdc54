#!/bin/bash
# dev: PAR skeleton experiments (EXCEL_DEV build): bit3 = stage half the rows, bit4 = one workgroup per CU (extra dynamic LDS)
for d in 0 8 16 7 15 23; do
  EXCEL_PAR_DBG=$d timeout 120 python bench.py --cpu-images 0 --ragged-images 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('dbg=$d', 'par_iterate', k['par_iterate'], 'step', d['ms_per_step'])"
done
