#!/bin/bash
# dev: same-box A/B of two builds of the library: bash tools_dev/ab.sh "cat1 cat2 ..." [reps]   (tools_dev/ab/base.so vs new.so)
CATS=${1:-"gemm_bf16x3 attn_accum attn_rowpass par_iterate"}
REPS=${2:-2}
for i in $(seq $REPS); do
  for v in base new; do
    EXCEL_AB_LIB=tools_dev/ab/$v.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$v', ' '.join('%s %.4f' % (c, k.get(c, 0)) for c in '$CATS'.split()), 'step', d['ms_per_step'])"
  done
done
