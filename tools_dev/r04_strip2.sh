#!/bin/bash
# round-4: producer / consumer strip kernel (EXCEL_STRIP2=1, dev library) vs the shipped kernel: parity tests first, then time
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04s}; mkdir -p $OUT
cp excel_amd/csrc/libexcel_hip.so /tmp/prod.so
cp tools_dev/ab/dev.so excel_amd/csrc/libexcel_hip.so
EXCEL_STRIP2=1 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "three_tiles or vit_b16_448_bf16x3_mode or outlier" > $OUT/pytest_strip2_a.log 2>&1; echo "strip2 pytest a rc $?" | tee $OUT/summary.txt
tail -n 3 $OUT/pytest_strip2_a.log
EXCEL_STRIP2=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "full_size or batch32 or lvc or golden or soak" > $OUT/pytest_strip2_b.log 2>&1; echo "strip2 pytest b rc $?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/pytest_strip2_b.log
cp /tmp/prod.so excel_amd/csrc/libexcel_hip.so
for rep in 1 2 3; do for v in 0 1; do
EXCEL_STRIP2=$v EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 8 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('strip2=$v', 'strip %.4f' % k['attn_accum'], 'rowpass %.4f' % k['attn_rowpass'], 'step', d['ms_per_step'], 'verify', d.get('verify', {}).get('label_agreement_mean'))" | tee -a $OUT/summary.txt
done; done
