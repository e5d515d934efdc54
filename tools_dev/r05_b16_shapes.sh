#!/bin/bash
# round 5: the B = 16 layer shapes (M = 12 560 at 448^2, 16 400 at 512^2) on the 8-wave tiles (EXCEL_BF_W4=0), each instance of the four-wave
# kernel (EXCEL_W4_NTM = 10 / 8 / 5) and the launcher's own choice; dev library
for M in ${1:-12560 16400}; do for shape in "2304 768" "768 768" "3072 768" "768 3072"; do
  echo -n "8-wave  "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_BF_W4=0 python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1
  for n in 10 8 5; do echo -n "w4-$n    "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_W4_NTM=$n python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1; done
  echo -n "chosen  "; EXCEL_AB_LIB=tools_dev/ab/dev.so python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1
done; done
