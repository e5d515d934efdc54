#!/bin/bash
# round-4: PAR step in plain grid order (dev library, EXCEL_PAR_DBG=32) against the shipped XCD-aware tile order (0): time per step, label
# agreement (the order is a permutation: results must not change), fabric fetch / write per launch.  (As first run the bits meant the
# opposite - 32 = remap, 64 = remap + 2-row bands, 0 = plain order: profiles/r04_ab_experiments.txt holds those lines.)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04n}; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for D in 0 32; do
  EXCEL_PAR_DBG=$D EXCEL_AB_LIB=tools_dev/ab/dev.so timeout 300 python tools_dev/ab_bench.py --cpu-images 0 --ragged-images 0 --steps 10 --warmup 3 2>$OUT/err_$D.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('par dbg=$D', 'par_iterate %.4f' % k['par_iterate'], 'step', d['ms_per_step'], 'hist_sha', str(d.get('hist_sha16', d.get('miou_synthetic'))))" | tee -a $OUT/par_xcd.txt
done
done
cd /tmp
for D in 0 32; do
  for C in FETCH_SIZE WRITE_SIZE; do
  EXCEL_PAR_DBG=$D EXCEL_AB_LIB=$GRAFT_REPO_ROOT/tools_dev/ab/dev.so rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/f$D -o f -- python $GRAFT_REPO_ROOT/tools_dev/ab_bench.py --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --no-kernel-timing > /dev/null 2> $OUT/f$D.err
  python - <<PY | tee -a $OUT/par_xcd.txt
import csv, glob, collections
f = glob.glob("$OUT/f$D/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    if r["Counter_Name"] == "$C": acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc:
    if "par_iterate" in k: print("par dbg=$D", k, n[k], "$C per launch MB (raw KB counter x 1024; FETCH still to be doubled): %.1f" % (acc[k] / n[k] * 1024 / 1e6))
PY
  rm -rf $OUT/f$D
  done
done
