#!/bin/bash
# round 6: launches made of two GEMM instances (tools_dev/ab/new.so) against uniform launches (base.so): layer shapes, then the whole step
for i in 1 2; do
  for v in base new; do
    echo "== $v"; EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/f16x2_bench.py 40 2>/dev/null | grep -E "qkv|fc1" | sed 's/f16x2\/split [0-9.]* us //'
  done
done
for i in 1 2 3; do
  for v in base new; do
    for w in seeded fp16; do
      EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/ab_bench.py --steps 20 --warmup 3 --cpu-images 0 --ragged-images 0 --fp16w-steps 0 --weights $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$v $w', d['value'], d['ms_per_step'], 'gemm', k['gemm_bf16x3'])"
    done
  done
done
