#!/bin/bash
# dev: same-box comparison of N builds of the library: bash tools_dev/abn.sh "cat1 cat2" reps base v1 v2 ...   (tools_dev/ab/<name>.so)
CATS=$1; REPS=$2; shift 2
for i in $(seq $REPS); do
  for v in "$@"; do
    EXCEL_AB_LIB=tools_dev/ab/$v.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$v', ' '.join('%s %.4f' % (c, k.get(c, 0)) for c in '$CATS'.split()), 'step', d['ms_per_step'])"
  done
done
