#!/bin/bash
# round 6: non-temporal epilogue stores of the split / q|k|v outputs in the four-wave GEMMs (tools_dev/ab/nt.so) against the shipped stores (base.so): whole step, both weight kinds, alternating
for i in 1 2 3; do
  for v in base nt; do
    for w in seeded fp16; do
      EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/ab_bench.py --steps 20 --warmup 3 --cpu-images 0 --ragged-images 0 --fp16w-steps 0 --weights $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$v $w', d['value'], d['ms_per_step'], 'gemm', k['gemm_bf16x3'], 'rowpass', k['attn_rowpass'], 'strip', k['attn_accum'], 'ln', k['layernorm'])"
    done
  done
done
