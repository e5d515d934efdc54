#!/bin/bash
# round 5: ablation arms of the four-wave GEMM (dev library tools_dev/ab/dev.so, EXCEL_W4_DBG: 1 no DMA, 2 no fragment reads, 4 no barrier,
# 8 no epilogue, 16 no MFMAs; sums combine) on the QKV and proj shapes; timing only
for shape in "25120 2304 768" "25120 768 768"; do
  for d in 0 8 9 10 24 15 1 2 4 0; do
    echo -n "dbg=$d "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_W4_DBG=$d python tools_dev/gemm_bench.py $shape 30 bf16x3_split 2>&1 | tail -1
  done
done
