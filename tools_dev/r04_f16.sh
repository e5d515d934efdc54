#!/bin/bash
# round-4 experiment: IEEE-half split planes (EXCEL_SPLIT_F16 build = tools_dev/ab/f16.so) vs the bf16 default: accuracy on the benign and
# the stress nets, then time
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/r04i; mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max 2>/dev/null | tee $OUT/cpu_quota.txt; nproc | tee -a $OUT/cpu_quota.txt; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6 | tee -a $OUT/cpu_quota.txt
for rep in 1 2; do
for L in prod f16; do
  if [ "$L" = "f16" ]; then export EXCEL_AB_LIB=tools_dev/ab/f16.so; else unset EXCEL_AB_LIB; fi
  timeout 300 python tools_dev/ab_bench.py --cpu-images 8 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$L.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$L', 'gemm %.4f' % k['gemm_bf16x3'], 'strip %.4f' % k['attn_accum'], 'rowpass %.4f' % k['attn_rowpass'], 'step', d['ms_per_step'], 'check', d.get('numerics_check'), 'verify', d.get('verify', {}).get('label_agreement_mean'))" | tee -a $OUT/f16_time.txt
done
done
unset EXCEL_AB_LIB
cp tools_dev/ab/f16.so excel_amd/csrc/libexcel_hip.so
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "outlier or vit_b16_448_bf16x3_mode or three_tiles or tiny_bf16x3 or baseline_batch16" > $OUT/pytest_f16.log 2>&1; echo "pytest f16 rc $?" | tee -a $OUT/f16_time.txt
grep -n "outlier net\|bf16x3:\|f32   :\|passed\|failed\|FAILED" $OUT/pytest_f16.log | tee -a $OUT/f16_time.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x > $OUT/pytest_f16_pipe.log 2>&1; echo "pytest f16 pipeline rc $?" | tee -a $OUT/f16_time.txt
tail -n 3 $OUT/pytest_f16_pipe.log
