#!/bin/bash
# usage: bash tools_dev/pmc.sh "<counters>" tag [kernel-name-filter]   (one rocprofv3 --pmc pass over a 2-step bench run; summary -> gpurun_out/pmc_<tag>/summary.txt)
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_$2; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -o p -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-kernel-timing > /dev/null 2> $OUT/err.txt
python - > $OUT/summary.txt <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = sorted({c for v in acc.values() for c in v})
print("kernel".ljust(46), "n".rjust(5), *[c[-24:].rjust(25) for c in names])
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:10]:
    nn = max(n[k].values())
    print(k.ljust(46), str(nn).rjust(5), *[f"{acc[k][c]/max(n[k][c],1):25.0f}" for c in names])
PY
find $OUT -name "*.csv" -delete
cat $OUT/summary.txt
