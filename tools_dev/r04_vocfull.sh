#!/bin/bash
# round-4: BASELINE configs[3]'s list size (10 582 variable-size images) from disk through the harness on one GPU (tools_dev/voc_full_run.py)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04h}; mkdir -p $OUT
N=${2:-10582}
export TMPDIR=/tmp
timeout 900 python tools_dev/voc_full_run.py write /tmp/voc_syn $N > $OUT/voc_write.log 2>&1; tail -n 1 $OUT/voc_write.log
timeout 900 python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1.json > $OUT/voc_run.log 2>&1; echo "run rc $?"; tail -n 4 $OUT/voc_run.log | cut -c1-1500
# second pass, page cache warm
timeout 900 python tools_dev/voc_full_run.py run /tmp/voc_syn $OUT/voc_full_n1_pass2.json > $OUT/voc_run2.log 2>&1; echo "run2 rc $?"; tail -n 1 $OUT/voc_run2.log | cut -c1-1500
