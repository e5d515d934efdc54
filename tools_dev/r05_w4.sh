#!/bin/bash
# round 5: the four-wave GEMM (gemm_w4.hip) against the 8-wave kernel, same box: parity tests, four layer shapes (dev library,
# EXCEL_BF_W4=0/1), then the pipeline
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm_bf16x3" 2>&1 | tail -5
for rep in 1 2; do for w in 0 1; do
  for shape in "25120 2304 768" "25120 768 768" "25120 3072 768" "25120 768 3072"; do
    echo -n "w4=$w "; EXCEL_AB_LIB=tools_dev/ab/dev.so EXCEL_BF_W4=$w python tools_dev/gemm_bench.py $shape 30 bf16x3_split 2>&1 | tail -1
  done
done; done
for rep in 1 2; do for w in 0 1; do
  echo -n "w4=$w "; EXCEL_BF_W4=$w EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print(' '.join('%s %.4f' % (c, k.get(c, 0)) for c in 'gemm_bf16x3 attn_accum attn_rowpass par_iterate'.split()), 'step', d['ms_per_step'], 'verify', d.get('verify'))"
done; done
