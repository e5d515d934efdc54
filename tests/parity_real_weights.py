#!/usr/bin/env python3
"""Operator parity check for the day real weights and real images are at hand (north-star: "CAMs within 1e-3 max abs of the reference
CPU path, VOC training-free mIoU reproduced"; reference: tools/infer_lam.py:63-176, logs/voc_train.log:114).

    python tests/parity_real_weights.py --model /path/ViT-B-16.pt --bpe_path /path/bpe_simple_vocab_16e6.txt.gz \
        --data_folder /data/VOC2012 --list_folder /data/voc_lists --infer_set val --n_images 64

runs the first N images of the list through
  (a) the CPU restatement of the reference (oracle/: torch-CPU ViT + numpy CAM / random walk / PAR, batch 1, fp32) and
  (b) the HIP path (excel_amd: the same batched ragged pipeline tools/infer_lam.py runs),
both built from THE SAME checkpoint (visual + text tower, shipped attribute bank), and prints ONE JSON line:
  cam_max_abs            max over images of |attr_maps_raw(HIP) - attr_maps_raw(CPU)|           gate 1e-3
  label_agreement_mean / _min   fraction of identical label pixels per image                    expect >= 0.999
  gemm_rung              the matrix-core mode ExCEL_model.check_numerics settles on for these weights (a published fp16 archive starts in f16x2 -> f32; fp32 weights: bf16x3 -> f16x3 -> f32) and the
                         CAM difference of every rung against exact fp32
  miou_hip / miou_cpu    training-free mIoU of both paths over the sample (and the histograms' L1 distance)
Exit status 0 when the CAM gate and the 99.9 % label gate hold, 3 otherwise.

This file lives under tests/ because it uses the oracle (test infrastructure; the product never imports it).  It is exercised in CI on a
tiny on-disk CLIP checkpoint + synthetic VOC tree (tests/test_gpu_parity_script.py); without real weights nothing here is a claim about
real-VOC numbers."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MEAN = np.array([123.675, 116.28, 103.53], np.float64)          # datasets/transforms.normalize_img
STD = np.array([58.395, 57.12, 57.375], np.float64)


def normalize_chw(img_u8):
    """datasets/transforms.normalize_img + HWC -> CHW on the host (double arithmetic, like the device kernel): uint8 [h,w,3] -> f32 [3,h,w]."""
    x = (img_u8.astype(np.float64) - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def main(argv=None):
    import torch
    import oracle
    from oracle import torch_cpu
    from oracle.vit import VitConfig
    from excel_amd import ops
    from excel_amd.datasets.loader import pack_samples
    from excel_amd.model.model_excel import ExCEL_model
    from excel_amd.pipeline import TrainingFreePipeline
    from excel_amd.tools import infer_lam
    from excel_amd.utils import evaluate

    ap = argparse.ArgumentParser(parents=[infer_lam.get_parser()], add_help=False, conflict_handler="resolve")
    ap.add_argument("--n_images", type=int, default=64)
    ap.add_argument("--cpu_threads", type=int, default=0, help="host threads of the CPU path (0 = torch default)")
    ap.add_argument("--cam_gate", type=float, default=1e-3)
    ap.add_argument("--label_gate", type=float, default=0.999)
    args = ap.parse_args(argv)
    if not args.data_folder:
        raise SystemExit("--data_folder / --list_folder (a VOC- or COCO-format tree) are required")
    assert torch.cuda.is_available(), "the HIP path needs an MI355X"
    device = torch.device("cuda", 0)
    if "coco" in args.dataset_name:
        from excel_amd.datasets import coco
        dataset = coco.CocoSegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.infer_set, stage="val")
    else:
        from excel_amd.datasets import voc
        dataset = voc.VOC12SegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.infer_set, stage="val")
    n = min(args.n_images, len(dataset))
    S, nc = args.resize_size, args.num_classes
    model = ExCEL_model(clip_model=args.model, embedding_dim=args.embedding_dim, in_channels=args.in_channels,
                        dataset_name=args.dataset_name, num_classes=nc, num_atrr_clusters=args.num_attri, json_file=args.attr_json,
                        img_size=S, mode=args.infer_set, device=device, gemm_mode=getattr(args, "gemm_mode", None),
                        **infer_lam.resolve_model_inputs(args))

    # ---- the numerics ladder on the first images (what infer_lam --gemm_check does before its loop)
    rb4 = pack_samples([dataset[i] for i in range(min(4, n))])
    inputs4 = ops.normalize_resize_u8_ragged(rb4.images.to(device), ops.RaggedPlan(rb4.hw, device), S)
    rung = model.check_numerics(inputs4, tol=5e-4, fallback=True)

    # ---- (b) HIP path: ragged batches, CAMs and labels kept per image
    pipe = TrainingFreePipeline(model, num_classes=nc, smax=dataset.max_k())
    hip_attr, hip_lab = [], []
    t0 = time.perf_counter()
    for s0 in range(0, n, args.batch_size):
        rb = pack_samples([dataset[i] for i in range(s0, min(s0 + args.batch_size, n))])
        plan = ops.RaggedPlan(rb.hw, device)
        labels, inter = pipe.run_batch_ragged(rb.images.to(device), plan, rb.cls.to(device), rb.labels.to(device), S=S, return_intermediates=True)
        for b in range(plan.B):
            hip_attr.append(inter["attr"][b].cpu().numpy())
            hip_lab.append(plan.label(labels, b).cpu().numpy())
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    hist_hip = pipe.hist.cpu().numpy()

    # ---- (a) CPU path from the same weights: the visual tower's own state dict (unprefixed keys, as the oracle reads them), the text
    # bank the model built from the checkpoint's text tower
    vis = model.encoder.visual
    cfg = VitConfig(width=vis.embed_dim, layers=vis.layers, heads=vis.num_heads, patch=vis.patch_size, out_dim=vis.output_dim,
                    input_resolution=vis.input_resolution, n_surgery=5)
    w = oracle.vit.reload_self_attn({k: np.asarray(torch.as_tensor(v).cpu().numpy(), np.float32) for k, v in vis.state_dict().items()},
                                    cfg, S // cfg.patch, args.infer_set)
    text_attr = model.text_attr.cpu().numpy()
    if args.cpu_threads:
        torch.set_num_threads(args.cpu_threads)
    vit = torch_cpu.TorchVit(w, cfg, S // cfg.patch)
    par = torch_cpu.TorchPAR((1, 2, 4, 8, 12, 24), 20)
    cam_err, agree, gts, cpu_lab = [], [], [], []
    t0 = time.perf_counter()
    for i in range(n):
        _, img, gt, cls = dataset[i]
        inputs = oracle.interp.bilinear_resize(normalize_chw(img)[None], S, S, align_corners=False)                   # infer_lam.py:74
        x, attn = vit.forward(inputs[0])
        f = x[None] / np.sqrt((x[None] * x[None]).sum(axis=1, keepdims=True, dtype=np.float32))                       # clip.py:353
        maps = oracle.cam.clip_feature_surgery(f.astype(np.float32), text_attr.T)[:, 1:, :nc - 1]                     # model_excel.py:57-58
        refined, cls_lst = oracle.aff.refine_cams_with_aff(maps[0], attn, cls, size=inputs.shape[2:], caa_thre=0.79)  # :93
        label, _ = oracle.aff.refine_cams_with_bkg_weclip(refined, inputs[0], cls_lst, par, gt.shape[-2:])            # :94
        cam_err.append(float(np.abs(hip_attr[i] - maps[0]).max()))
        agree.append(float((hip_lab[i] == label[0]).mean()))
        cpu_lab.append(label[0].astype(np.int16))
        gts.append(np.asarray(gt).astype(np.int16))
    t_cpu = time.perf_counter() - t0
    hist_cpu = oracle.evaluate.hist_of(gts, cpu_lab, nc)
    out = {
        "images": n, "resize_size": S, "dataset": args.dataset_name, "checkpoint": str(args.model),
        "cam_max_abs": max(cam_err), "cam_max_abs_mean_over_images": float(np.mean(cam_err)), "cam_gate": args.cam_gate,
        "label_agreement_mean": float(np.mean(agree)), "label_agreement_min": float(np.min(agree)), "label_gate": args.label_gate,
        "gemm_rung": {"started_in": rung["mode_before"], "settled_on": rung["mode_after"],
                      "cam_diff_vs_exact_fp32": {m: d for m, d in rung["ladder"]}, "tol": rung["tol"]},
        "miou_hip": float(evaluate.scores_from_hist(torch.from_numpy(hist_hip))["miou"]),
        "miou_cpu": float(evaluate.scores_from_hist(torch.from_numpy(hist_cpu))["miou"]),
        "hist_l1": int(np.abs(hist_hip - hist_cpu).sum()), "scored_pixels": int(hist_cpu.sum()),
        "seconds_hip": round(t_hip, 3), "seconds_cpu": round(t_cpu, 3), "cpu_threads": torch.get_num_threads(),
    }
    out["ok"] = bool(out["cam_max_abs"] <= args.cam_gate and out["label_agreement_mean"] >= args.label_gate)
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    sys.exit(0 if main()["ok"] else 3)
