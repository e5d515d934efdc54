#!/bin/bash
# round 5: is the four-wave GEMM's epilogue bound by the fabric (all CUs storing at once) or by the CU?  Full kernel vs no-epilogue arm
# (EXCEL_W4_DBG=8) on grids of 24 / 96 / 237 / 711 tiles, plain fp32 and split output; the 8-wave kernel beside it (EXCEL_BF_W4=0, its
# no-epilogue arm is EXCEL_BF_DBG=8)
for mode in bf16x3 bf16x3_split; do
for shape in "2560 768 768" "10240 768 768" "25120 768 768" "25120 2304 768"; do
  for d in 0 8; do
    echo -n "w4 dbg=$d "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_BF_TILE=320 EXCEL_W4_DBG=$d python tools_dev/gemm_bench.py $shape 30 $mode 2>&1 | tail -1
  done
  for d in 0 8; do
    echo -n "w8 dbg=$d "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_BF_TILE=320 EXCEL_BF_W4=0 EXCEL_BF_DBG=$d python tools_dev/gemm_bench.py $shape 30 $mode 2>&1 | tail -1
  done
done; done
