#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/r04j; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_all.log 2>&1; echo "pytest all rc $?" | tee $OUT/summary.txt
tail -n 3 $OUT/pytest_all.log; grep "outlier net" $OUT/pytest_all.log | tee -a $OUT/summary.txt
for rep in 1 2; do for m in bf16x3 f16x3; do
EXCEL_GEMM_MODE=$m timeout 300 python bench.py --cpu-images 8 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$m.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$m', d['dtype'][:8], 'gemm %.4f' % k.get('gemm_bf16x3',0), 'strip %.4f' % k['attn_accum'], 'rowpass %.4f' % k['attn_rowpass'], 'step', d['ms_per_step'], 'value', d['value'], 'check', d.get('numerics_check',{}).get('max_abs_diff'), 'verify', d.get('verify', {}).get('label_agreement_mean'))" | tee -a $OUT/summary.txt
done; done
