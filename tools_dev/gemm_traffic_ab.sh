#!/bin/bash
# dev: FETCH_SIZE per launch of the wide-N GEMM shapes, uniform 320-row tiles (EXCEL_BF_UNIFORM=1) vs mixed-height tiles, dev library
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/gemm_traffic_ab; rm -rf $OUT; mkdir -p $OUT
for name in "qkv 25120 2304 768" "fc1 25120 3072 768" "qkv16 12560 2304 768"; do
  set -- $name
  for U in 1 0; do
    if [ "$U" = "1" ]; then export EXCEL_BF_UNIFORM=1; else unset EXCEL_BF_UNIFORM; fi
    (cd /tmp && EXCEL_AB_LIB=$REPO/tools_dev/ab/dev.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$1_$U -o g -- python $REPO/tools_dev/gemm_bench.py $2 $3 $4 10 bf16x3_split > $OUT/$1_$U.txt 2> $OUT/$1_$U.err)
    python - <<PY
import csv, glob
f = glob.glob("$OUT/$1_$U/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_bf16x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$1 uniform=$U fetch x2 per launch MB %.1f  (A+W = %.1f MB)" % (2 * sum(v) / len(v) * 1024 / 1e6, ($2 * $4 + $3 * $4) * 4 / 1e6), open("$OUT/$1_$U.txt").read().strip())
PY
  done
done
unset EXCEL_BF_UNIFORM
for name in "qkv 25120 2304 768" "fc1 25120 3072 768"; do
  set -- $name
  for rep in 1 2; do for U in 1 0; do
    if [ "$U" = "1" ]; then export EXCEL_BF_UNIFORM=1; else unset EXCEL_BF_UNIFORM; fi
    echo "$1 uniform=$U $(EXCEL_AB_LIB=$REPO/tools_dev/ab/dev.so python $REPO/tools_dev/gemm_bench.py $2 $3 $4 30 bf16x3_split 2>/dev/null)"
  done; done
done
