#!/bin/bash
# round 5: (a) PAR: centre read in front of the DMA issue (parh) vs parg1; (b) four-wave GEMM: first-round stagger of D us (dev arm EXCEL_W4_STAGGER)
bash tools_dev/abn.sh "par_iterate gemm_bf16x3" 2 parg1 parh
for rep in 1 2; do for D in 0 6 12 18 24; do
  for shape in "25120 2304 768" "25120 3072 768"; do
    for mode in bf16x3_split bf16x3; do
    echo -n "stagger $D "; EXCEL_W4_STAGGER=$D EXCEL_AB_LIB=tools_dev/ab/w4dev.so python tools_dev/gemm_bench.py $shape 30 $mode 2>&1 | tail -1
    done
  done
done; done
for rep in 1 2; do for D in 0 12 18; do
  echo -n "pipeline stagger $D: "; EXCEL_W4_STAGGER=$D EXCEL_AB_LIB=tools_dev/ab/w4dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print(' '.join('%s %.4f' % (c, k.get(c, 0)) for c in 'gemm_bf16x3 par_iterate'.split()), 'step', d['ms_per_step'])"
done; done
