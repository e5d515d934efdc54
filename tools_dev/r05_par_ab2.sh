#!/bin/bash
# round 5: PAR step, same box: base = committed head; pare1 = statistics loads one plane ahead; parg1 = + counted wait in front of the mask-plane barrier (stores stay in flight)
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -x -q -k "par or PAR or ragged" 2>&1 | tail -4
bash tools_dev/abn.sh "par_iterate par_affinity gemm_bf16x3" 3 base pare1 parg1
