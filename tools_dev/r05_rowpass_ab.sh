#!/bin/bash
# round 5: flash row pass with the next step's key fragments read in the P.V operand batch (rowpk = working tree (packed softmax in the flash row pass)) vs w4gelu (head 68421ad), same box
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -x -q -k "vit or attn or attention or batch or golden or lvc or configs" 2>&1 | tail -3
bash tools_dev/abn.sh "attn_rowpass attn_accum gemm_bf16x3 par_iterate" 3 w4gelu rowpk
