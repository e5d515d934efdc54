#!/bin/bash
# dev: same-box A/B of the compact-weight two-product GEMM between tools_dev/ab/base.so and new.so (alternating processes)
for i in 1 2 3; do
  for v in base new; do
    echo "== $v"; EXCEL_AB_LIB=tools_dev/ab/$v.so F16X2_ONLY=half python tools_dev/f16x2_bench.py 40 2>/dev/null | sed 's/bit-identical.*//'
  done
done
