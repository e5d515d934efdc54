"""Training-free LAM evaluation harness - mirror of tools/infer_lam.py (build_validation :63-128, validate :130-176).

Differences from the reference, all deliberate:
  * batches of B > 1 images run through the device-resident TrainingFreePipeline (the reference is batch 1 with
    per-class host round trips); `--api_path` runs the reference's per-image call sequence instead.  Real data (`--data_folder`:
    every image has its own size, and the path refines and scores at that size, :74,:94) and `--ragged true` synthetic data run
    as RAGGED batches: a background decode pool (`--num_workers` threads; `--decode processes` = the reference's DataLoader worker
    processes, :167) fills a pinned staging ring, a copy stream moves the batches to the device ahead of the compute stream, the device resizes every image to the network size and runs the size-dependent half (up-sampling, PAR, arg-max) over packed
    planes through a tile map - one launch per stage for any mix of sizes (pipeline.run_batch_ragged);
  * ranks take images r, r+R, ... exactly like :166, accumulate a device-side [nc,nc] int64 confusion matrix and
    exchange it ONCE with an all_gather over RCCL (the reference scores each shard separately and never aggregates);
  * weights: `--model` is a CLIP checkpoint path or a model name found under --clip_root / $EXCEL_CLIP_ROOT / ~/.cache/clip
    (clip.load, model/model_excel.py:25); the class + background prompts are encoded with the checkpoint's text tower
    (model/model_excel.py:31-33; needs CLIP's BPE merges file: --bpe_path / $EXCEL_BPE_VOCAB).  `--model_path` = the trained
    decoder checkpoint, required only with `--training_free false` (the reference loads and then ignores it otherwise, :150-152);
  * `--synthetic N` (no --data_folder) feeds seeded synthetic samples; seeded random weights are used ONLY in that mode and only
    when no checkpoint can be resolved (logged).  `--data_folder` without a resolvable checkpoint is an error.
Launch: python -m torch.distributed.run --nproc-per-node R -m excel_amd.tools.infer_lam --synthetic 64 ...
"""
import argparse
import logging
import os
import time

import numpy as np
import torch
import torch.distributed as dist

VOC_CLASSES = ["_background_", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
               "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def _bool(x):
    return x.lower() in ["true", "1", "yes"]


def get_parser():
    p = argparse.ArgumentParser()
    # flags kept from the reference (tools/infer_lam.py:30-60)
    p.add_argument("--model_path", default=None, type=str)
    p.add_argument("--model", default="ExCEL_ViT-B/16", type=str)
    p.add_argument("--dataset_name", default="pascal_voc", type=str)
    p.add_argument("--attr_json", default=None, type=str)
    p.add_argument("--num_attri", default=112, type=int)
    p.add_argument("--embedding_dim", default=256, type=int)
    p.add_argument("--in_channels", default=768, type=int)
    p.add_argument("--resize_size", default=448, type=int)
    p.add_argument("--infer_set", default="train", type=str)
    p.add_argument("--training_free", default=True, type=_bool)
    p.add_argument("--data_folder", default=None, type=str, help="VOC2012 root (JPEGImages/, SegmentationClassAug/): real data instead of --synthetic")
    p.add_argument("--list_folder", default=None, type=str, help="directory with <infer_set>.txt and cls_labels_onehot.npy")
    p.add_argument("--u8_input", default=False, type=_bool, help="feed decoded uint8 HWC images and normalise on the device")
    p.add_argument("--crf_post", default=False, type=_bool, help="write the per-image logits records (:116-119, api path) and run the DenseCRF stage over them (:179-237)")
    p.add_argument("--logits_dir", default="./logits", type=str)
    p.add_argument("--segs_crf_rgb_dir", default=None, type=str, help="colour-coded CRF label images (tools/infer_lam.py:228); default: not written")
    p.add_argument("--num_classes", default=21, type=int)
    p.add_argument("--ignore_index", default=255, type=int)
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
    p.add_argument("--backend", default="nccl")
    # additions
    p.add_argument("--batch_size", default=32, type=int)
    p.add_argument("--synthetic", default=64, type=int, help="number of seeded synthetic samples")
    p.add_argument("--seed", default=1234, type=int)
    p.add_argument("--api_path", default=False, type=_bool, help="per-image reference call sequence instead of the batched pipeline")
    p.add_argument("--ragged", default=False, type=_bool, help="synthetic samples with VOC-like, per-image sizes (uint8 images), fed as ragged batches like real data")
    p.add_argument("--num_workers", default=-1, type=int,
                   help="background decoders of the ragged path (the reference's DataLoader uses 2 worker processes, :167); -1 = from the CPUs "
                        "this rank may really use (affinity and the container's cgroup quota, shared by the ranks of the node): at most 16")
    p.add_argument("--decode", default="threads", choices=["threads", "processes"],
                   help="threads: a decode thread pool in this process (Pillow releases the GIL; default); processes: forked DataLoader "
                        "workers like the reference - after such workers exit, host-side GPU event waits of this process were measured "
                        "at ~350 ms each on ROCm 7.2, so the default avoids fork")
    p.add_argument("--clip_root", default=None, type=str, help="directory holding the published CLIP archive (ViT-B-16.pt); default $EXCEL_CLIP_ROOT, ~/.cache/clip")
    p.add_argument("--bpe_path", default=None, type=str, help="CLIP's bpe_simple_vocab_16e6.txt.gz (default $EXCEL_BPE_VOCAB)")
    p.add_argument("--gemm_mode", default=None, type=str, help="auto (default: f16x2 when every GEMM weight is fp16-valued - every published CLIP archive -, else bf16x3) | bf16x3 | f16x3 | f16x2 | f32")
    p.add_argument("--gemm_check", default=True, type=_bool,
                   help="before the loop, run the first few images in the chosen fast mode AND in exact fp32 and compare the CAMs; above "
                        "--gemm_check_tol the run moves down the ladder bf16x3 -> f16x3 (f16x2 on fp16-valued weights) -> f32 (ill-conditioned weights; "
                        "ExCEL_model.check_numerics)")
    p.add_argument("--gemm_check_tol", default=5e-4, type=float)
    p.add_argument("--cpu_affinity", default="auto", choices=["auto", "off"],
                   help="auto: with several ranks on the node every rank pins itself (decode pool included) to its own share of the host cores")
    p.add_argument("--json_out", default=None, type=str, help="rank 0 writes a one-line JSON record of the run here (rate, ranks, per-rank mass)")
    return p


# ------------------------------------------------------------------ sharding + the one collective
def shard_indices(n, rank, world):
    """Subset(np.arange(i, len, n_gpus)) (tools/infer_lam.py:166)."""
    return np.arange(rank, n, world)


def host_cpu_budget():
    """CPUs this process may really use: the affinity mask capped by the container's cgroup-v2 quota (`cpu.max`; the GPU boxes of this
    project report 256 logical CPUs and grant 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def default_decode_workers(local_world=1):
    """Decode threads per rank: this rank's share of the CPU budget minus the launching thread, between 2 and 16 (measured: 16 threads are
    best for one rank on 16 granted CPUs, 8 per rank for 8 ranks)."""
    return max(2, min(16, host_cpu_budget() // max(local_world, 1) - (1 if local_world > 1 else 0)))


def pin_rank_to_cores(local_rank, local_world):
    """Give local rank r of R its own contiguous share of the CPUs this process may run on (os.sched_setaffinity): the decode pool and
    the launch thread of a rank then stay off the other ranks' cores (8 ranks x 16 decode threads + 8 launch threads on one host,
    BASELINE configs[3]).  -> the core set, or None when there are fewer cores than ranks / the platform has no affinity call."""
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < local_world:
        return None
    per = len(cores) // local_world
    mine = set(cores[local_rank * per:(local_rank + 1) * per])
    os.sched_setaffinity(0, mine)
    return mine


def gather_hists(hist, group=None):
    """all_gather of the per-rank [nc,nc] int64 confusion matrices -> ([R,nc,nc], summed [nc,nc]).
    One small message per rank (3.5 KB VOC / 52 KB COCO): latency-bound, issued once per evaluation.
    With a process group the collective always runs - also over a single rank (RCCL on one GPU: the same call path as N > 1, which is
    what the 1-GPU hardware tests and `bench.py --gpus 1` exercise); without one the matrix is returned as it is."""
    if not (dist.is_available() and dist.is_initialized()):
        return hist[None], hist
    world = dist.get_world_size(group)
    parts = [torch.empty_like(hist) for _ in range(world)]
    dist.all_gather(parts, hist.contiguous(), group=group)
    stacked = torch.stack(parts, 0)
    return stacked, stacked.sum(0)


def format_scores_table(score, cat_list, metric_names=("confusion", "precision", "recall", "iou")):
    """The table of utils/pyutils.py:37-58 (format_tabs_multi_metircs) without the texttable dependency."""
    rows = [["Class"] + list(metric_names)]
    vals = np.array([list(score[m].values()) for m in metric_names], np.float64)
    for i in range(vals.shape[1]):
        rows.append([cat_list[i] if i < len(cat_list) else str(i)] + [f"{v:.4f}" for v in vals[:, i]])
    rows.append(["average_metrics"] + [f"{v:.4f}" for v in vals.mean(1)])
    widths = [max(len(r[c]) for r in rows) for c in range(len(rows[0]))]
    line = "+" + "+".join("-" * (w + 2) for w in widths) + "+"
    out = [line]
    for i, r in enumerate(rows):
        out.append("|" + "|".join(" " + r[c].ljust(widths[c]) + " " for c in range(len(r))) + "|")
        if i == 0 or i == len(rows) - 2:
            out.append(line)
    out.append(line)
    return "\n".join(out)


# ------------------------------------------------------------------ the loop
def _check_present_classes(batches, smax):
    """The compacted class list of an image has `smax` slots (the kernels clamp to it): an image with MORE present classes than the data
    set's max_k() promised would silently lose the extra ones - refuse it instead, on the host, from the batch's own one-hot rows."""
    for rb in batches:
        k = rb.cls.sum(1)
        if int(k.max()) > smax:
            b = int(k.argmax())
            raise RuntimeError(f"{rb.names[b]}: {int(k[b])} present classes, the pipeline was built for at most {smax} (dataset.max_k()): "
                               "build TrainingFreePipeline with a larger smax")
        yield rb


def build_validation(model=None, par=None, dataset=None, indices=None, device="cuda", args=None, pipe=None):
    """-> (hist [nc,nc] int64 on device, images processed, seconds).  Mirrors :63-128."""
    from ..pipeline import TrainingFreePipeline
    from ..utils import evaluate
    from ..utils.affutils import refine_cams_with_aff, refine_cams_with_bkg_weclip
    from .. import ops
    S = args.resize_size
    on_gpu = torch.device(device).type == "cuda"
    if pipe is None:
        pipe = TrainingFreePipeline(model, num_classes=args.num_classes, dilations=par.dilations, num_iter=par.num_iter,
                                    caa_thre=0.79, smax=dataset.max_k())
    hist = torch.zeros((args.num_classes, args.num_classes), dtype=torch.int64, device=device)
    t0 = time.time()
    nimg = 0
    training_free = bool(getattr(args, "training_free", True))
    per_image = args.api_path or not training_free
    if getattr(args, "ragged_batches", False) and not per_image:
        # every sample at its own size: decode in background workers, everything else on the device, one launch per stage
        from ..datasets.loader import ragged_batches
        from ..utils import imutils
        pipe.hist = hist
        keep = bool(getattr(args, "crf_post", False))
        nw = int(getattr(args, "num_workers", -1))
        if nw < 0:
            nw = default_decode_workers(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", 1))))
        if getattr(args, "decode", "threads") == "processes" and nw > 0:    # the reference's mechanism (DataLoader worker processes, :167)
            batches = ragged_batches(dataset, indices, args.batch_size, num_workers=nw, pin_memory=False)
        else:                                                               # default: a thread pool (datasets/loader.threaded_batches)
            from ..datasets.loader import threaded_batches
            batches = threaded_batches(dataset, indices, args.batch_size, num_threads=max(nw, 1))
        batches = _check_present_classes(batches, pipe.smax)
        if on_gpu:
            from ..datasets.loader import DeviceFeeder
            feed = DeviceFeeder(batches, device)              # H2D on a copy stream, a few batches ahead
        else:                                                  # (control-flow tests: a stub pipeline on CPU tensors)
            feed = ((rb.names, ops.RaggedPlan(rb.hw, None), rb.images, rb.cls, rb.labels) for rb in batches)
        for names, plan, images, cls_t, labels_t in feed:
            out = pipe.run_batch_ragged(images, plan, cls_t, labels_t, S=S, return_intermediates=keep)
            if keep:                                                                        # :116-119 record for the CRF stage
                inter = out[1]
                cls_idx, ncls = inter["cls_idx"].cpu().numpy(), inter["ncls"].cpu().numpy()
                for b, name in enumerate(names):
                    k = min(int(ncls[b]), pipe.smax)            # (ncls <= smax is enforced below; the record stays consistent anyway)
                    imutils.save_logits(args.logits_dir, name, plan.planes(inter["cams"], b, pipe.smax + 1)[:k + 1], cls_idx[b, :k].astype(np.int64),
                                        run_token=getattr(args, "run_token", None))
            nimg += len(names)
        if on_gpu:
            torch.cuda.synchronize()
        return pipe.hist, nimg, time.time() - t0
    bs = 1 if per_image else args.batch_size
    for s in range(0, len(indices), bs):
        names, imgs, gts, cls = dataset.batch(indices[s:s + bs])
        inputs = torch.from_numpy(imgs).to(device, non_blocking=True)
        if inputs.dtype == torch.uint8:                                                     # decoded images: normalise on the device
            inputs = ops.normalize_img_u8(inputs)                                           # datasets/voc.py:115-116
        if inputs.shape[-2:] != (S, S):
            inputs = ops.bilinear_resize(inputs, S, S, align_corners=False)                 # :74
        cls_labels = torch.from_numpy(cls).to(device, non_blocking=True)
        gt_dev = torch.from_numpy(gts).to(device, non_blocking=True)
        if per_image:
            if training_free:
                _, _, attr_maps_raw, attn_weights, attn_pred = model(inputs)                # :79
            else:
                # optimised-LAM regime (:84-85, :91): needs the caller's decoder as model.feature_head
                from ..utils.camutils import cure_attr_map_flip
                _, _, _, attn_weights, attn_pred = model(inputs, n_attn_out=6)              # :79
                if attn_pred is None:
                    raise RuntimeError("--training_free false needs model.feature_head (the learned decoder, SURVEY 8f #2)")
                attr_maps_raw = cure_attr_map_flip(model, inputs)                           # :85
            for i, attr_map in enumerate(attr_maps_raw):                                    # :88
                seg_attn = None if training_free else attn_pred[i][None]                    # :91-92
                refined, cls_lst = refine_cams_with_aff(attr_map, attn_weights[:, i], cls_labels[i], size=inputs.shape[2:],
                                                        seg_attn=seg_attn, caa_thre=0.79)   # :93
                labels, normed = refine_cams_with_bkg_weclip(refined, inputs[i], cls_lst, par, gts.shape[-2:])   # :94
                if getattr(args, "crf_post", False):                                         # :116-119 record for the CRF stage
                    from ..utils import imutils
                    imutils.save_logits(args.logits_dir, str(names[i]), normed, cls_lst, run_token=getattr(args, "run_token", None))
                hist = evaluate.hist_from_labels([gt_dev[i]], [labels[0]], args.num_classes, device, hist)
        else:
            pipe.hist = hist
            pipe.run_batch(inputs, cls_labels, gt_dev)
            hist = pipe.hist
        nimg += len(imgs)
    torch.cuda.synchronize()
    return hist, nimg, time.time() - t0


def resolve_model_inputs(args):
    """What ExCEL_model is built from (tools/infer_lam.py:144-162, model/model_excel.py:25-34) -> keyword arguments:
      state_dict          the CLIP checkpoint --model denotes (path, or model name under --clip_root / $EXCEL_CLIP_ROOT / ~/.cache/clip)
      tokenizer           CLIP's BPE tokenizer (the class/background prompts are encoded by the checkpoint's text tower)
      decoder_state_dict  --model_path (trained head, `module.` prefixes stripped, positional embedding skipped: :153-160)
                          when --training_free false
    Seeded random weights + random unit-norm text features are returned ONLY for --synthetic runs without a resolvable
    checkpoint; real data without weights raises."""
    from .. import clip
    from . import synthetic
    kw = {}
    ckpt = clip.clip.find_checkpoint(args.model, getattr(args, "clip_root", None))
    if ckpt is not None:
        sd = clip.clip.read_checkpoint(ckpt)
        kw["state_dict"] = sd
        if not any(k in sd for k in ("text_projection", "visual.proj")) and "proj" not in sd:
            raise RuntimeError(f"{ckpt}: not a CLIP checkpoint (no visual.proj / text_projection)")
        if "text_projection" not in sd:
            raise RuntimeError(f"{ckpt}: the checkpoint has no text tower (text_projection ...): the class / background prompts of "
                               "model/model_excel.py:31-33 cannot be encoded")
        from ..clip import bpe
        kw["tokenizer"] = bpe.BPETokenizer(getattr(args, "bpe_path", None))            # raises FileNotFoundError with the remedy
        logging.info(f"CLIP weights: {ckpt}")
    elif getattr(args, "data_folder", None):
        raise RuntimeError(f"--data_folder given but no CLIP checkpoint found for --model {args.model!r}: pass a checkpoint path, or put "
                           "the published archive (ViT-B-16.pt) under --clip_root / $EXCEL_CLIP_ROOT / ~/.cache/clip. "
                           "Refusing to score real data with random weights.")
    else:
        T = 45 if args.num_classes <= 21 else 103
        kw["state_dict"] = synthetic.make_vit_state_dict(seed=0)
        kw["text_features"] = synthetic.make_text_features(T)
        logging.warning("SYNTHETIC run: seeded random ViT-B/16 weights and random text features (no CLIP checkpoint resolved); "
                        "the mIoU below is a throughput / plumbing check, not a result")
    if not bool(getattr(args, "training_free", True)):
        if not args.model_path or not os.path.isfile(args.model_path):
            raise RuntimeError("--training_free false needs --model_path (the trained decoder checkpoint, tools/infer_lam.py:150-160)")
        trained = torch.load(args.model_path, map_location="cpu")
        dec = {}
        for k, v in trained.items():                                                        # :153-160
            k = k.replace("module.", "")
            if "encoder.visual.positional_embedding" not in k and k.startswith(("decoder_fts_fuse.", "decoder.")):
                dec[k] = v
        if not dec:
            raise RuntimeError(f"{args.model_path}: no decoder_fts_fuse.* / decoder.* tensors")
        kw["decoder_state_dict"] = dec
    return kw


def crf_proc(args, rank=0, world=1, device="cuda"):
    """tools/infer_lam.py:179-237: DenseCRF (iter 10, pos_xy_std 1, pos_w 3, bi_xy_std 67, bi_rgb_std 3, bi_w 4, :191-198) over the logits
    records of the main loop, label = pad(keys_gt + 1)[argmax] (:225-226), colour-coded PNG (:228), scores against the ground truth
    (:233).  The reference fans the images out over CPU processes (joblib); here every rank takes names r, r+R, ... through the
    device-side mean field (excel_dcrf_inference) and the confusion matrices are all-gathered like the main loop's.
    -> (score dict, summed hist)"""
    from PIL import Image
    from ..utils import evaluate, imutils
    from ..utils.dcrf import DenseCRF
    from .. import ops
    with open(os.path.join(args.list_folder, args.infer_set) + ".txt") as f:
        name_list = [x for x in f.read().split("\n") if x]                                  # :182-184
    images_path = os.path.join(args.data_folder, "JPEGImages")
    labels_path = os.path.join(args.data_folder, "SegmentationClassAug")
    post = DenseCRF(iter_max=10, pos_xy_std=1, pos_w=3, bi_xy_std=67, bi_rgb_std=3, bi_w=4)  # :191-198
    hist = torch.zeros((args.num_classes, args.num_classes), dtype=torch.int64, device=device)
    for i in shard_indices(len(name_list), rank, world):
        name = name_list[i]
        rec = os.path.join(args.logits_dir, name + ".npy")
        # records carry the token of the run that wrote them (validate() draws one and shares it with every rank): no dependence on
        # file-system time stamps or clocks.  A caller that runs build_validation + crf_proc without a token accepts any record.
        token = getattr(args, "run_token", None)
        if not os.path.isfile(rec) or (token is not None and imutils.logits_run_token(rec) != token):
            raise RuntimeError(f"crf_proc: {rec} was not written by this run (missing or stale): the CRF stage scores the records of the "
                               "main loop (tools/infer_lam.py:116-119), never those of an earlier one")
        lams, keys = imutils.load_logits(rec)                                               # :203-206
        image = np.asarray(Image.open(os.path.join(images_path, name + ".jpg")).convert("RGB")).astype(np.uint8)   # :209-210, :220
        if "test" in args.infer_set:
            label = image[:, :, 0]                                                          # :213-214
        else:
            label = np.array(Image.open(os.path.join(labels_path, name + ".png")))          # :216
        prob = post(torch.from_numpy(image).to(device), torch.from_numpy(np.asarray(lams, np.float32)).to(device))   # :221
        pred = prob.argmax(0)                                                               # :222
        keys_t = torch.from_numpy(np.pad(np.asarray(keys) + 1, (1, 0), mode="constant")).to(device)   # :225
        pred_crf = keys_t[pred].to(torch.uint8)                                             # :226
        if getattr(args, "segs_crf_rgb_dir", None):
            os.makedirs(args.segs_crf_rgb_dir, exist_ok=True)
            Image.fromarray(imutils.encode_cmap(pred_crf.cpu().numpy())).save(os.path.join(args.segs_crf_rgb_dir, name + ".png"))   # :228
        hist = ops.confusion_accumulate(torch.from_numpy(np.array(label, dtype=np.uint8)).to(device), pred_crf, args.num_classes, hist)
    _, total = gather_hists(hist)
    return evaluate.scores_from_hist(total), total                                          # :233


def _gemm_self_check(model, dataset, idx, args, device, world):
    """ExCEL_model.check_numerics (the one implementation of the bf16x3 -> f16x3 -> f32 ladder) on the first (up to 4) samples of this
    rank's shard, resized like the loop resizes them (:74).  With several ranks every rung's verdict is shared (all-reduce MAX of the
    difference): every rank runs the same mode, and a rank whose shard is EMPTY (fewer images than ranks) still joins every collective
    with a difference of 0 instead of leaving the others waiting."""
    from .. import ops
    S = args.resize_size
    take = [int(i) for i in idx[:4]]
    inputs = None
    if take:
        first = dataset[take[0]][1]
        if first.dtype == np.uint8:                                                         # decoded images of their own sizes
            from ..datasets.loader import pack_samples
            rb = pack_samples([dataset[i] for i in take])
            inputs = ops.normalize_resize_u8_ragged(rb.images.to(device), ops.RaggedPlan(rb.hw, device), S)
        else:
            _, imgs, _, _ = dataset.batch(take)
            inputs = torch.from_numpy(imgs).to(device)
            if inputs.shape[-2:] != (S, S):
                inputs = ops.bilinear_resize(inputs, S, S, align_corners=False)

    def share(diff):
        if world > 1 and dist.is_initialized():
            t = torch.tensor([diff], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return diff

    tol = float(getattr(args, "gemm_check_tol", 5e-4))
    out = model.check_numerics(inputs, tol=tol, fallback=True, reduce=share)
    before, after, ladder = out["mode_before"], out["mode_after"], out["ladder"]
    if after != before:
        logging.warning(f"gemm self-check: CAMs of the {before} mode differ from exact fp32 by {ladder[0][1]:.2e} (> {tol:.1e}) on these weights: "
                        f"the run continues in {after} ({', '.join(f'{m} {d:.2e}' for m, d in ladder)})")
    elif ladder:
        logging.info(f"gemm self-check: {before} vs exact fp32 CAM max-abs difference {ladder[0][1]:.2e} (tolerance {tol:.1e})")
    return out


def validate(args=None, dataset=None, pipe=None):
    """tools/infer_lam.py:130-176.  `dataset` / `pipe` are injection points for the multi-rank control-flow tests (a stub pipeline on
    CPU tensors over gloo): with `pipe` given no model is built and the device is pipe.device; the product path passes neither."""
    from ..utils import evaluate
    from ..utils.PAR import PAR
    from . import synthetic
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if pipe is None:
        torch.cuda.set_device(args.local_rank)
        device = torch.device("cuda", args.local_rank)
    else:
        device = torch.device(pipe.device)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=args.backend)                                       # :133
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if pipe is None and getattr(args, "cpu_affinity", "auto") == "auto" and local_world > 1:
        cores = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", args.local_rank)), local_world)
        if cores and rank == 0:
            logging.info(f"rank 0 pinned to {len(cores)} of the host's cores (every rank takes its own share; --cpu_affinity off disables)")
    # one token per evaluation, the same on every rank (rank 0 draws it): stamps the CRF records this run writes
    import uuid
    tok = [uuid.uuid4().hex]
    if world > 1:
        dist.broadcast_object_list(tok, src=0)
    args.run_token = tok[0]
    if dataset is not None:
        args.ragged_batches = True
    elif getattr(args, "data_folder", None):
        # :156-163: every image has its own size.  The reference's harness always builds the VOC data set (:132); a COCO tree
        # (--dataset_name ms_coco, BASELINE configs[4]) gets the COCO reader of datasets/coco.py here
        if "coco" in args.dataset_name:
            from ..datasets import coco
            dataset = coco.CocoSegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.infer_set, stage="val")
        else:
            from ..datasets import voc
            dataset = voc.VOC12SegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.infer_set, stage="val")
        args.ragged_batches = True
    else:
        args.ragged_batches = bool(getattr(args, "ragged", False))
        dataset = synthetic.SyntheticSegDataset(args.synthetic, (args.resize_size, args.resize_size), num_classes=args.num_classes,
                                                seed=args.seed, u8_images=getattr(args, "u8_input", False), ragged=args.ragged_batches)
    model = None
    if pipe is None:
        from ..model.model_excel import ExCEL_model
        model = ExCEL_model(clip_model=args.model, embedding_dim=args.embedding_dim, in_channels=args.in_channels,
                            dataset_name=args.dataset_name, num_classes=args.num_classes, num_atrr_clusters=args.num_attri,
                            json_file=args.attr_json, img_size=args.resize_size, mode=args.infer_set, device=device,
                            gemm_mode=getattr(args, "gemm_mode", None), **resolve_model_inputs(args))
    # everything built so far (torch, the weights, the data set index) lives for the whole run: move it out of the cyclic collector's
    # reach, or every full collection walks it again - 40-70 ms of host time each (measured in bench.py, where one landed in a timed step)
    import gc
    gc.collect()
    gc.freeze()
    par = PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])                                  # :168
    idx = shard_indices(len(dataset), rank, world)                                          # :166
    if model is not None and getattr(args, "gemm_check", True):              # every rank joins, also one with an empty shard
        validate.last_gemm_check = _gemm_self_check(model, dataset, idx, args, device, world)
    hist, nimg, secs = build_validation(model, par, dataset, idx, device, args, pipe=pipe)
    validate.last_model = model                                                             # handle for callers / tests
    per_rank, total = gather_hists(hist)
    validate.last_per_rank = per_rank
    score = evaluate.scores_from_hist(total)
    cat_list = VOC_CLASSES                                                                  # :123
    if "voc" not in args.dataset_name:
        from ..datasets import coco
        cat_list = coco.class_list
    if rank == 0:
        logging.info(f"Training_free:{args.training_free}, LAM_score:")
        logging.info("\n" + format_scores_table(score, cat_list))
        logging.info(f"mIoU {score['miou'] * 100:.3f}  images {int(nimg) * world}  ({nimg / secs:.1f} img/s/rank)")
        if getattr(args, "json_out", None):
            # one self-checking record per run (tools_dev/scale.sh): wall time, rate and every rank's scored-pixel mass
            import json
            with open(args.json_out, "w") as f:
                json.dump({"world": world, "rccl_ranks": int(dist.get_world_size()) if world > 1 else 1, "images_rank0": int(nimg),
                           "images_total": int(len(dataset)), "seconds_rank0": round(secs, 3), "images_per_s_rank0": round(nimg / secs, 2),
                           "images_per_s_job": round(len(dataset) / secs, 2), "miou": float(score["miou"]),
                           "per_rank_hist_mass": [int(x) for x in per_rank.reshape(per_rank.shape[0], -1).sum(1).tolist()],
                           "hist_total": [[int(v) for v in row] for row in total.cpu().tolist()],       # the gathered confusion matrix itself
                           "batch_size": args.batch_size, "ragged": bool(args.ragged_batches), "resize_size": args.resize_size}, f)
    if getattr(args, "crf_post", False) and getattr(args, "data_folder", None):             # :173-174
        crf_score, crf_total = crf_proc(args, rank, world, device)
        validate.last_crf = (crf_score, crf_total)
        if rank == 0:
            logging.info("crf_seg_score:")
            logging.info("\n" + format_scores_table(crf_score, cat_list))
    return score, total


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    validate(get_parser().parse_args())
