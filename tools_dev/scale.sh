#!/bin/bash
# Multi-GPU scaling run for the first 8-GPU lease: bench.py at N = 1,2,4,8 (BASELINE configs[2] shape, weak scaling) and the
# training-free harness over 10 582 synthetic VOC-shaped samples (configs[3]) / the COCO-shaped config (configs[4]), one JSON line
# per run under gpurun_out/scale/.  Usage: bash tools_dev/scale.sh [max_gpus]
set -u
MAXN=${1:-8}
OUT=$PWD/gpurun_out/scale
mkdir -p "$OUT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29541
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXN" ] && break
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + N)) \
        bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  fi
  tail -c 400 "$OUT/bench_n$N.json"; echo
done
# configs[3]: VOC train_aug-sized evaluation (10 582 images with VOC-like per-image sizes - ragged batches of 32, background decode
# workers -, rank r takes r, r+R, ...; one RCCL all-gather of the confusion matrix); one JSON record per N with wall time, img/s and
# the gathered per-rank mass (a rank that did not run shows as a zero)
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXN" ] && break
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 20 + N)) \
      -m excel_amd.tools.infer_lam --synthetic 10582 --ragged true --batch_size 32 --num_workers 8 --resize_size 448 \
      --json_out "$OUT/infer_lam_voc_n$N.json" > "$OUT/infer_lam_voc_n$N.log" 2>&1
  cat "$OUT/infer_lam_voc_n$N.json"; echo
done
# configs[4]: COCO-shaped (81 classes, 512x512, batch 16)
N=$MAXN
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((PORT + 40)) \
    -m excel_amd.tools.infer_lam --synthetic 4999 --batch_size 16 --resize_size 512 --num_classes 81 --dataset_name ms_coco --num_attri 224 \
    --json_out "$OUT/infer_lam_coco_n$N.json" > "$OUT/infer_lam_coco_n$N.log" 2>&1
cat "$OUT/infer_lam_coco_n$N.json"; echo
