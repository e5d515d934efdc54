#!/bin/bash
# round-4: GEMM mixed-height tiles: parity at the benchmark's shapes, then same-box A/B (dev library: EXCEL_BF_UNIFORM=1 = uniform tiles)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04e}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm or vit_b16 or outlier or batch16" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm rc $?" | tee -a $OUT/gemm_time.txt
tail -n 4 $OUT/pytest_gemm.log
for rep in 1 2 3; do
for U in 1 0; do
  if [ "$U" = "1" ]; then export EXCEL_BF_UNIFORM=1; else unset EXCEL_BF_UNIFORM; fi
  EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$U.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('uniform $U', 'gemm %.4f' % k['gemm_bf16x3'], 'strip %.4f' % k['attn_accum'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'])" | tee -a $OUT/gemm_time.txt
done
done
unset EXCEL_BF_UNIFORM
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $OUT/gemm_time.txt
tail -n 4 $OUT/pytest_all.log
