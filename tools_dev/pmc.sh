#!/bin/bash
# usage: bash tools_dev/pmc.sh "<counters>" tag
export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_$2; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -o p -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-kernel-timing > /dev/null 2> $OUT/err.txt
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
import glob
f = glob.glob("$OUT/*counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == list(acc[k].keys())[0]: n[k] += 1
names = sorted({c for v in acc.values() for c in v})
print("kernel".ljust(46), "n".rjust(5), *[c[-22:].rjust(23) for c in names])
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:9]:
    print(k.ljust(46), str(n[k]).rjust(5), *[f"{acc[k][c]/max(n[k],1):23.0f}" for c in names])
PY
rm -f $OUT/*.csv
