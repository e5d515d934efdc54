#!/bin/bash
# dev (EXCEL_DEV build): same-box A/B of EXCEL_BF_DBG variants on the layer shapes: bash tools_dev/gemm_ab.sh "0 8"
for rep in 1 2; do for d in $1; do
  for shape in "25120 2304 768" "25120 768 768" "25120 3072 768" "25120 768 3072"; do
    echo -n "dbg=$d "; EXCEL_BF_DBG=$d python tools_dev/gemm_bench.py $shape 30 bf16x3_split 2>&1 | tail -1
  done
done; done
