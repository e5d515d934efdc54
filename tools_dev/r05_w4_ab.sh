#!/bin/bash
# round 5: two dev builds of the library (tools_dev/ab/base.so = committed head, dev.so = working tree), same box: GEMM parity tests on the
# working tree, the four layer shapes in three output modes, then the pipeline
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm_bf16x3" 2>&1 | tail -2
for rep in 1 2; do for v in base dev; do
  for shape in "25120 2304 768" "25120 768 768" "25120 3072 768" "25120 768 3072"; do
    for mode in bf16x3_split bf16x3; do
    echo -n "$v "; EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/gemm_bench.py $shape 30 $mode 2>&1 | tail -1
    done
  done
done; done
for rep in 1 2 3; do for v in base dev; do
  echo -n "$v "; EXCEL_AB_LIB=tools_dev/ab/$v.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print(' '.join('%s %.4f' % (c, k.get(c, 0)) for c in 'gemm_bf16x3 attn_accum attn_rowpass par_iterate'.split()), 'step', d['ms_per_step'])"
done; done
