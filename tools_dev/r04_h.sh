#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/r04h; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_all.log 2>&1; echo "pytest all rc $?" | tee $OUT/summary.txt
tail -n 3 $OUT/pytest_all.log; grep "outlier net" $OUT/pytest_all.log | tee -a $OUT/summary.txt
bash tools_dev/par_ablate2.sh $OUT/par_arms.txt
for v in 0 3; do for rep in 1 2; do
EXCEL_STRIP_VAR=$v EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('strip var $v', 'strip %.4f' % k['attn_accum'], 'step', d['ms_per_step'])" | tee -a $OUT/summary.txt
done; done
# decode ceiling: glibc malloc thresholds (no mmap / munmap churn for the ~560 KB image arrays)
timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads 8 --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
MALLOC_MMAP_THRESHOLD_=268435456 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_ARENA_MAX=4 timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads 8 --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
MALLOC_MMAP_THRESHOLD_=268435456 MALLOC_TRIM_THRESHOLD_=1073741824 timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads 8 --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
timeout 600 python tools_dev/decode_ceiling.py --procs 16 --threads 8 --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
timeout 600 python tools_dev/decode_ceiling.py --procs 32 --threads 4 --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
cat $OUT/decode_ceiling.jsonl
