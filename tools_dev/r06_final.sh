#!/bin/bash
# round 6: the final evidence set on ONE box (run through gpurun from the repo root):  bash tools_dev/r06_final.sh <git head>
H=${1:-unknown}
mkdir -p gpurun_out/r06f
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06f/gputests_tail.txt
bash profiles/collect.sh r06f_main $H > gpurun_out/r06f/collect_main.log 2>&1
BENCH_ARGS="--weights fp16" bash profiles/collect.sh r06f_fp16 $H > gpurun_out/r06f/collect_fp16.log 2>&1
python bench.py > gpurun_out/r06f/bench_n1_second_line.json 2> /dev/null
python tools_dev/configs_bench.py 10 > gpurun_out/r06f/configs_bench.txt 2>&1
python tools_dev/configs_bench.py 10 fp16 > gpurun_out/r06f/configs_bench_fp16.txt 2>&1
python tools_dev/f16x2_bench.py 30 > gpurun_out/r06f/f16x2_shapes.txt 2>&1
python tools_dev/par_stream_probe.py > gpurun_out/r06f/par_stream_probe.txt 2>&1
SOAK_MODE=f32 python tools_dev/soak.py 32 > gpurun_out/r06f/soak.txt 2>&1
bash tools_dev/r06_voc_full.sh r06f/voc > gpurun_out/r06f/voc_summary.txt 2>&1
cat gpurun_out/r06f/gputests_tail.txt gpurun_out/r06f/configs_bench.txt gpurun_out/r06f/configs_bench_fp16.txt gpurun_out/r06f/voc_summary.txt; tail -3 gpurun_out/r06f/soak.txt
