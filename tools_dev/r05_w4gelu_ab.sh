#!/bin/bash
# round 5: four-wave GEMM epilogue: stage-wise packed QuickGELU + predicate-free copy for tiles inside the matrix (w4gelu = working tree) vs w4res (head 87ffb98)
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -x -q -k "gemm or vit or batch or configs or split" 2>&1 | tail -3
for rep in 1 2; do for v in w4res w4gelu; do
  for shape in "25120 2304 768" "25120 3072 768"; do
    echo -n "$v "; EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/gemm_bench.py $shape 30 bf16x3_split 2>&1 | tail -1
  done
  echo -n "$v gelu "; EXCEL_AB_LIB=tools_dev/ab/$v.so python tools_dev/gemm_bench.py 25120 3072 768 30 bf16x3_gelu_split 2>&1 | tail -1
done; done
bash tools_dev/abn.sh "gemm_bf16x3 attn_accum par_iterate" 3 w4res w4gelu
