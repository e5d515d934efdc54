#!/bin/bash
# round 6: the step as 1 / 2 concurrent sub-batches and as the two-stream pipeline (ViT+CAM of batch i+1 over PAR of batch i), same box, alternating
for i in 1 2; do
  for a in "--split 1" "--split 2" "--overlap 1" "--split 1 --weights fp16" "--split 2 --weights fp16" "--overlap 1 --weights fp16"; do
    python bench.py --steps 20 --warmup 3 --cpu-images 0 --ragged-images 0 --power-seconds 0 --fp16w-steps 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a', d['value'], d['ms_per_step'], d['config']['gemm_mode'])"
  done
done
