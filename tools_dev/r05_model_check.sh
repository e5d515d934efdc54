#!/bin/bash
# round 5: the launcher's choice after the partial-round refinement of the four-wave time model (dev2.so), every layer shape of B = 16 / 32
for M in 12560 16400 25120; do for shape in "2304 768" "768 768" "3072 768" "768 3072"; do
  echo -n "chosen  "; EXCEL_AB_LIB=tools_dev/ab/dev2.so python tools_dev/gemm_bench.py $M $shape 30 bf16x3_split 2>&1 | tail -1
done; done
python tools_dev/configs_bench.py 2>/dev/null
