#!/bin/bash
# round-4: one rank's shard of BASELINE configs[4] (COCO val2014 / 8 = 5 063 images, 80 classes, 512^2, batch 16) from an on-disk COCO-format
# tree through the harness on one GPU (tools_dev/voc_full_run.py)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04u}; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools_dev/voc_full_run.py write /tmp/coco_syn 5063 coco > $OUT/coco_write.log 2>&1; tail -n 1 $OUT/coco_write.log
timeout 900 python tools_dev/voc_full_run.py run /tmp/coco_syn $OUT/coco_shard_n1.json > $OUT/coco_run.log 2>&1; echo "run rc $?"; tail -n 3 $OUT/coco_run.log | cut -c1-1600
