#!/bin/bash
# round-4: what the GEMM epilogue costs per tile, contended (a full round of 252 tiles) and uncontended (36 tiles); dev arms EXCEL_BF_DBG:
# 8 = no epilogue, 16 = epilogue without its global stores, 32 = without its LDS transpose, 48 = neither (math + addressing only)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04j}; mkdir -p $OUT
export TMPDIR=/tmp
export EXCEL_AB_LIB=tools_dev/ab/dev.so
export EXCEL_BF_TILE=320
for rep in 1 2; do
for S in "1280 2304 768 bf16x3_split" "8960 2304 768 bf16x3_split" "1280 768 768 bf16x3" "26880 768 768 bf16x3"; do
  set -- $S
  for D in 0 8 16 32 48; do
    EXCEL_BF_DBG=$D timeout 120 python tools_dev/gemm_bench.py $1 $2 $3 40 $4 2>/dev/null | sed "s/^/dbg=$D  /" | tee -a $OUT/epi2.txt
  done
done
done
