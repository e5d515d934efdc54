#!/bin/bash
# round 5: four-wave GEMM, residual rows requested one row tile ahead (w4res = working tree) vs stripk (head d6c03b7), same box
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
for rep in 1 2; do for v in stripk w4res; do
  for shape in "25120 768 768" "25120 768 3072"; do
    echo -n "$v res "; EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/gemm_bench.py $shape 30 bf16x3_res 2>&1 | tail -1
  done
done; done
bash tools_dev/abn.sh "gemm_bf16x3 attn_accum par_iterate" 3 stripk w4res
