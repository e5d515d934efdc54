#!/bin/bash
# round-4: finer k-loop arms of the 16x16x32 GEMM (dev library built from a scratch copy with three more EXCEL_BF_DBG bits: 128 no per-step
# barrier, 256 no B fragment reads, 512 no A fragment reads after row tile 0); all without epilogue (8) and without DMA (2) unless noted
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04ae}; mkdir -p $OUT
export TMPDIR=/tmp EXCEL_AB_LIB=tools_dev/ab/dev.so
for rep in 1 2; do
for D in 8 10 138 266 522 778 906 12; do
  EXCEL_BF_DBG=$D timeout 120 python tools_dev/gemm_bench.py 25120 2304 768 40 bf16x3_split 2>/dev/null | sed "s/^/dbg=$D  /" | tee -a $OUT/arms2.txt
done
done
