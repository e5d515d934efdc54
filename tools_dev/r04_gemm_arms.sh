#!/bin/bash
# round-4: ablation arms of the 16x16x32 GEMM (dev library): EXCEL_BF_DBG 0 full, 8 no epilogue, 2 no DMA after the first tile, 10 both,
# 4 MFMA-only loop, 12 MFMA-only without epilogue; full-size layer shapes and one uncontended 36-tile launch
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04z}; mkdir -p $OUT
export TMPDIR=/tmp EXCEL_AB_LIB=tools_dev/ab/dev.so
for rep in 1 2; do
for S in "25120 2304 768 bf16x3_split" "25120 768 3072 bf16x3" "25120 768 768 bf16x3" "25120 3072 768 bf16x3_split"; do
  set -- $S
  for D in 0 8 2 10 4 12; do
    EXCEL_BF_DBG=$D timeout 120 python tools_dev/gemm_bench.py $1 $2 $3 40 $4 2>/dev/null | sed "s/^/dbg=$D  /" | tee -a $OUT/arms.txt
  done
done
done
