#!/bin/bash
# round 5: PAR step variants, same box: base = committed head; pard3 = mask-phase read-ahead 3 only; pare1/2/3 = statistics loads one plane
# ahead (in front of the LDS-DMA pieces) + mask-phase read-ahead 1/2/3
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -x -q -k "par or PAR or ragged" 2>&1 | tail -15
bash tools_dev/abn.sh "par_iterate par_affinity gemm_bf16x3" 2 base pard3 pare1 pare2 pare3
