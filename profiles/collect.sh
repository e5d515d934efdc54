#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: bench line, kernel-trace stats, the two fabric-traffic PMC passes and two
# occupancy/busy PMC passes.  Usage: [BENCH_ARGS="--weights fp16"] bash profiles/collect.sh <tag> [git head]     -> gpurun_out/<tag>/...   (copy what should be judged to profiles/)
# BENCH_ARGS: extra bench.py arguments of every pass (round 6: "--weights fp16" profiles the f16x2 step on checkpoint-like weights)
set -u
BENCH_ARGS=${BENCH_ARGS:-}
TAG=${1:-prof}
HEAD=${2:-unknown}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
export TMPDIR=/tmp
python bench.py $BENCH_ARGS > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cd /tmp
# the SAME command under the tracer (kt_bench.json = its bench line; its roofline.avg_launch_ms vs the category lines of kernel_stats.txt)
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python "$REPO/bench.py" $BENCH_ARGS --cpu-images 0 --ragged-images 0 --power-seconds 0 > "$OUT/kt_bench.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name '*.db' | head -1)
[ -n "$DB" ] && python "$REPO/profiles/summarize_rocpd.py" "$DB" > "$OUT/kernel_stats.txt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- python "$REPO/bench.py" $BENCH_ARGS --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --power-seconds 0 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- python "$REPO/bench.py" $BENCH_ARGS --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --power-seconds 0 > /dev/null 2> "$OUT/write.err"
F=$(find "$OUT" -name 'fetch_counter_collection.csv' | head -1); W=$(find "$OUT" -name 'write_counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python "$REPO/profiles/summarize_pmc.py" "$F" "$W" "$HEAD" > "$OUT/hbm_traffic.json"
# matrix-core / VALU / LDS busy and wave-state split per kernel (SQ counters; GRBM_GUI_ACTIVE = kernel cycles x 8 XCDs)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT" -o busy1 -- python "$REPO/bench.py" $BENCH_ARGS --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --power-seconds 0 --no-kernel-timing > /dev/null 2> "$OUT/busy1.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d "$OUT" -o busy2 -- python "$REPO/bench.py" $BENCH_ARGS --steps 2 --warmup 1 --cpu-images 0 --ragged-images 0 --power-seconds 0 --no-kernel-timing > /dev/null 2> "$OUT/busy2.err"
B1=$(find "$OUT" -name 'busy1_counter_collection.csv' | head -1); B2=$(find "$OUT" -name 'busy2_counter_collection.csv' | head -1)
[ -n "$B1" ] && [ -n "$B2" ] && python "$REPO/profiles/summarize_busy.py" "$B1" "$B2" > "$OUT/pipe_busy.txt"
find "$OUT/kt" -name "*.db" -delete; rm -f "$OUT"/*_counter_collection.csv "$OUT"/*_kernel_trace.csv "$OUT"/*/*_counter_collection.csv "$OUT"/*/*_kernel_trace.csv 2>/dev/null
ls -la "$OUT"; cat "$OUT/bench_n1.json"; head -24 "$OUT/kernel_stats.txt"; cat "$OUT/pipe_busy.txt"
