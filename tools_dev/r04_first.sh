#!/bin/bash
# round-4 first GPU call: parity tier, headline bench, multi-GPU plumbing probes, host decode ceiling
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r04a/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/pytest_gpu.log
tail -5 gpurun_out/r04a/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench_n1.json 2> gpurun_out/r04a/bench_n1.err; echo "bench rc $?"
tail -c 600 gpurun_out/r04a/bench_n1.json; echo
# bare multi-GPU command on a 1-GPU box: must refuse with the GPU count, not with a launcher message
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04a/bench_gpus2.out 2>&1; echo "bench --gpus 2 rc $?" >> gpurun_out/r04a/bench_gpus2.out
tail -2 gpurun_out/r04a/bench_gpus2.out
# the driver-shaped launch at nproc 1 (RCCL init + all-gather + all-reduce through torchrun)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools_dev/rccl_smoke.py > gpurun_out/r04a/rccl_smoke.log 2>&1; echo "rccl rc $?" >> gpurun_out/r04a/rccl_smoke.log
tail -2 gpurun_out/r04a/rccl_smoke.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-images 0 --ragged-images 0 > gpurun_out/r04a/bench_torchrun_n1.json 2> gpurun_out/r04a/bench_torchrun_n1.err; echo "torchrun bench rc $?"
# host decode ceiling of configs[3] at 8 ranks
for aff in "" "--no-affinity"; do
  for th in 8 16; do
    timeout 600 python tools_dev/decode_ceiling.py --procs 8 --threads $th --images 512 --passes 3 $aff >> gpurun_out/r04a/decode_ceiling.jsonl 2>> gpurun_out/r04a/decode_ceiling.err
  done
done
timeout 300 python tools_dev/decode_ceiling.py --procs 1 --threads 16 --images 512 --passes 2 --no-affinity >> gpurun_out/r04a/decode_ceiling.jsonl 2>> gpurun_out/r04a/decode_ceiling.err
cat gpurun_out/r04a/decode_ceiling.jsonl
