#!/bin/bash
# round-4: GEMM on v_mfma_f32_16x16x32 (tools_dev/ab/dev.so = the production build) against the 32x32x16 build (tools_dev/ab/old.so):
# parity tests first, then four layer shapes standalone, then the pipeline, alternating
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04w}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or vit_b16 or outlier or batch16 or vit_tiny or batch32" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm rc $?" | tee -a $OUT/mfma1616.txt
tail -n 3 $OUT/pytest_gemm.log
for rep in 1 2; do
for LIB in old dev; do
  export EXCEL_AB_LIB=tools_dev/ab/$LIB.so
  for S in "25120 2304 768 bf16x3_split" "25120 768 768 bf16x3" "25120 3072 768 bf16x3_split" "25120 768 3072 bf16x3"; do
    set -- $S
    timeout 120 python tools_dev/gemm_bench.py $1 $2 $3 40 $4 2>/dev/null | sed "s/^/$LIB  /" | tee -a $OUT/mfma1616.txt
  done
done
done
for rep in 1 2 3; do
for LIB in old dev; do
  EXCEL_AB_LIB=tools_dev/ab/$LIB.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$LIB.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('pipeline $LIB', 'gemm %.4f' % k['gemm_bf16x3'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'])" | tee -a $OUT/mfma1616.txt
done
done
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $OUT/mfma1616.txt
tail -n 3 $OUT/pytest_all.log
