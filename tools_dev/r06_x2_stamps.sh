#!/bin/bash
# dev: cycle stamps and ablation arms of the compact-weight two-product GEMM (tools_dev/ab/dev.so = EXCEL_DEV build)
export EXCEL_AB_LIB=tools_dev/ab/dev.so
for dbg in 128 136 143; do
  for shape in "25120 2304 768" "25120 768 3072"; do
    EXCEL_W4_DBG=$dbg python tools_dev/w4_stamps.py $shape 0 x2 2>/dev/null | cut -c1-900
  done
done
echo "---- arms (us per launch): 0 full, 1 no DMA, 2 no fragment reads, 4 no barrier, 8 no epilogue, 15 MFMA only(+epilogue)"
for dbg in 0 1 2 4 8 9 10 15; do
  echo "dbg $dbg"; EXCEL_W4_DBG=$dbg F16X2_ONLY=half python tools_dev/f16x2_bench.py 30 2>/dev/null | head -4 | sed 's/bit-identical.*//'
done
