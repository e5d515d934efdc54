#!/bin/bash
# round 6: BASELINE configs[3]'s whole list on one GPU, from fp32-stored weights (bf16x3) and from fp16-stored weights like the published
# archive (the harness starts in f16x2).  bash tools_dev/r06_voc_full.sh <out dir under gpurun_out>
OUT=gpurun_out/${1:-r06_voc}
mkdir -p $OUT
python tools_dev/voc_full_run.py write /tmp/voc_syn 10582 > $OUT/write.log 2>&1
python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1_bf16x3.json > $OUT/run_bf16x3.log 2>&1
python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1_f16x2.json --model /tmp/voc_syn/fp16/ViT-B-16.pt > $OUT/run_f16x2.log 2>&1
python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1_bf16x3_pass2.json > $OUT/run_bf16x3_2.log 2>&1
python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1_f16x2_pass2.json --model /tmp/voc_syn/fp16/ViT-B-16.pt > $OUT/run_f16x2_2.log 2>&1
for f in $OUT/voc_full_n1_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', d.get('images_per_s_job'), d.get('seconds_rank0'), d.get('every_pixel_scored_once'), (d.get('gemm_check') or {}).get('mode_after'), d.get('miou'))"; done
