#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench line, kernel-trace stats and the two PMC passes.
# Usage: bash profiles/collect.sh <tag>     -> gpurun_out/<tag>/...
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp
python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cd /tmp
# the SAME command under the tracer (kt_bench.json = its bench line; its roofline.avg_launch_ms vs the category lines of kernel_stats.txt)
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python "$REPO/bench.py" > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name '*.db' | head -1)
[ -n "$DB" ] && python "$REPO/profiles/summarize_rocpd.py" "$DB" > "$OUT/kernel_stats.txt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 --cpu-images 0 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 --cpu-images 0 > /dev/null 2> "$OUT/write.err"
F=$(find "$OUT" -name 'fetch_counter_collection.csv' | head -1); W=$(find "$OUT" -name 'write_counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python "$REPO/profiles/summarize_pmc.py" "$F" "$W" > "$OUT/hbm_traffic.json"
find "$OUT/kt" -name "*.db" -delete; rm -f "$OUT"/*_counter_collection.csv "$OUT"/*_kernel_trace.csv 2>/dev/null
ls -la "$OUT"; cat "$OUT/bench_n1.json"; head -20 "$OUT/kernel_stats.txt"
