#!/bin/bash
# round-4: attention strip kernel variants (EXCEL_STRIP_VAR, dev build tools_dev/ab/dev.so): time per step, then the parity tests on the variant
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04c}; mkdir -p $OUT
VARS=${2:-"0 1 3 7"}
export TMPDIR=/tmp
for rep in 1 2; do
for V in $VARS; do
  EXCEL_STRIP_VAR=$V EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$V.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('var $V', 'strip %.4f' % k['attn_accum'], 'rowpass %.4f' % k['attn_rowpass'], 'gemm %.4f' % k['gemm_bf16x3'], 'step', d['ms_per_step'])" | tee -a $OUT/strip_time.txt
done
done
# parity of the variants: the shipped tests against the dev library with the variant forced
cp tools_dev/ab/dev.so excel_amd/csrc/libexcel_hip.so
for V in $VARS; do
  [ "$V" = "0" ] && continue
  EXCEL_STRIP_VAR=$V timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -x -q -k "vit or attn or golden or pipeline or batch32 or batch16 or coco or outlier or soak" > $OUT/pytest_var$V.log 2>&1; echo "var $V pytest rc $?" | tee -a $OUT/strip_time.txt
  tail -n 3 $OUT/pytest_var$V.log
done
