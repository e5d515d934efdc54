#!/bin/bash
# round-4: staggered LDS-DMA slots per wave inside a row tile's MFMAs (dev library, EXCEL_BF_DBG=64) against all-at-the-head (0)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04aa}; mkdir -p $OUT
export TMPDIR=/tmp EXCEL_AB_LIB=tools_dev/ab/dev.so
for rep in 1 2; do
for D in 0 64; do
for S in "25120 2304 768 bf16x3_split" "25120 768 3072 bf16x3" "25120 768 768 bf16x3" "25120 3072 768 bf16x3_split"; do
  set -- $S
  EXCEL_BF_DBG=$D timeout 120 python tools_dev/gemm_bench.py $1 $2 $3 40 $4 2>/dev/null | sed "s/^/dbg=$D  /" | tee -a $OUT/slots.txt
done
done
done
for rep in 1 2; do
for D in 0 64; do
  EXCEL_BF_DBG=$D timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('pipeline dbg=$D', 'gemm %.4f' % k['gemm_bf16x3'], 'step', d['ms_per_step'], 'miou', d.get('miou_synthetic'))" | tee -a $OUT/slots.txt
done
done
