#!/bin/bash
# dev: PAR guide-kernel ablations (library built with EXCEL_DEV=1): bit0 no weight accumulation, bit1 no exp/finalise, bit2 no mask phase
for d in 0 1 2 4 3 7; do
  EXCEL_PAR_DBG=$d timeout 120 python bench.py --cpu-images 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('dbg=$d', 'par_iterate', k['par_iterate'], 'step', d['ms_per_step'])"
done
