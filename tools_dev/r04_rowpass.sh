#!/bin/bash
# round-4: row-pass tail order (groups per chunk) - time per step and fabric bytes per launch, dev build (tools_dev/ab/dev.so)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_ops.py > $OUT/pytest_rest.log 2>&1; echo "pytest(rest) rc $?" | tee -a $OUT/pytest_rest.log
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "outlier or batch16 or narrow or bench_shapes" > $OUT/pytest_new.log 2>&1; echo "pytest(new) rc $?" | tee -a $OUT/pytest_new.log
tail -3 $OUT/pytest_rest.log $OUT/pytest_new.log
for rep in 1 2; do
for G in 0 1 2 4 8 16; do
  EXCEL_ROWPASS_GRP=$G EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('grp $G', 'rowpass %.4f' % k['attn_rowpass'], 'strip %.4f' % k['attn_accum'], 'step', d['ms_per_step'])" | tee -a $OUT/rowpass_time.txt
done
done
cd /tmp
for G in 0 2 4 8; do
  EXCEL_ROWPASS_GRP=$G EXCEL_AB_LIB=$GRAFT_REPO_ROOT/tools_dev/ab/dev.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f$G -o f -- python $GRAFT_REPO_ROOT/tools_dev/ab_bench.py --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --no-kernel-timing > /dev/null 2> $OUT/f$G.err
  python - <<PY | tee -a $OUT/rowpass_fetch.txt
import csv, glob, collections
f = glob.glob("$OUT/f$G/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if r["Counter_Name"] == "FETCH_SIZE": acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc:
    if "rowpass" in k or "strip" in k: print("grp $G", k, n[k], "FETCH_SIZE x2 per launch MB: %.1f" % (2 * acc[k] / n[k] * 1024 / 1e6))
PY
  rm -rf $OUT/f$G
done
cd $GRAFT_REPO_ROOT
for th in 8 16; do
  timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads $th --images 512 --passes 40 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
done
timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads 16 --images 512 --passes 40 --no-affinity >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
timeout 600 python tools_dev/decode_ceiling.py --procs 4 --threads 16 --images 512 --passes 20 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
timeout 600 python tools_dev/decode_ceiling.py --procs 2 --threads 16 --images 512 --passes 10 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
timeout 600 python tools_dev/decode_ceiling.py --procs 1 --threads 16 --images 512 --passes 8 >> $OUT/decode_ceiling.jsonl 2>> $OUT/decode_ceiling.err
cat $OUT/decode_ceiling.jsonl
