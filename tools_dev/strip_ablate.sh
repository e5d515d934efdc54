#!/bin/bash
# dev: attention strip kernel ablations (needs a library built with EXCEL_DEV=1)
for d in 0 1 2 4 3 8; do
  EXCEL_STRIP_DBG=$d timeout 120 python bench.py --cpu-images 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('dbg=$d', 'accum', k['attn_accum'], 'rowpass', k['attn_rowpass'], 'step', d['ms_per_step'])"
done
