#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$PWD/gpurun_out/${1:-r04f}; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1; echo "pytest all rc $?" | tee $OUT/summary.txt
tail -n 5 $OUT/pytest_all.log; grep -n "outlier net" $OUT/pytest_all.log | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc $?" | tee -a $OUT/summary.txt
python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print(d['value'], d['ms_per_step'], d['numerics_check'], d['roofline']['frac'], d['kernel_ms_per_step'])" | tee -a $OUT/summary.txt
