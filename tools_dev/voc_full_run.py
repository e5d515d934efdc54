"""BASELINE configs[3] on its own terms on ONE GPU: the whole `train_aug`-sized list (10 582 variable-size images) from an on-disk
PASCAL-VOC-format tree through `excel_amd.tools.infer_lam --data_folder` (JPEG / PNG decode in the background, ragged batches of 32,
on-disk CLIP-format checkpoint + BPE file), one JSON record of the run.

There are no VOC images and no ViT-B-16.pt in this image: the tree is synthetic (excel_amd.tools.synthetic's VOC-like sizes and class
counts, JPEG quality 90), the checkpoint is a seeded random ViT-B/16 visual tower + a small text tower in the published key layout.  The
mIoU of the record is plumbing; images, sizes, file formats, batch shape and the code path are the real harness's.

  step 1 (no GPU, forks):  python tools_dev/voc_full_run.py write /tmp/voc_syn 10582        (... write /tmp/coco_syn 5063 coco: a COCO-format tree)
  step 2:                  python tools_dev/voc_full_run.py run /tmp/voc_syn gpurun_out/<tag>/voc_full_n1.json
"""
import gzip
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

MERGES = ["a n", "t h", "i n", "e r", "o n", "r e", "th e</w>", "c l", "o r", "in g</w>", "a t", "e n", "o u", "a r", "e s</w>", "cl e", "an d</w>"]


def _write_slice(job):
    root, lo, hi, n, seed, coco = job
    from PIL import Image
    from excel_amd.tools.synthetic import SyntheticSegDataset
    from excel_amd.utils import imutils
    nc = 81 if coco else 21
    ds = SyntheticSegDataset(n, num_classes=nc, seed=seed, ragged=True)
    palette = imutils.colormap().flatten().tolist()
    out, npix = {}, 0
    for i in range(lo, hi):
        _, img, gt, cls = ds[i]
        # (datasets/coco.py: JPEGImages/val/COCO_val2014_<12 digits>.jpg, SegmentationClass/val/<12 digits>.png)
        name = ("COCO_val2014_%012d" % i) if coco else ("2008_%06d" % i)
        img_dir = os.path.join(root, "JPEGImages", "val") if coco else os.path.join(root, "JPEGImages")
        lab_path = os.path.join(root, "SegmentationClass", "val", name[13:] + ".png") if coco else os.path.join(root, "SegmentationClassAug", name + ".png")
        Image.fromarray(img).save(os.path.join(img_dir, name + ".jpg"), quality=90)
        im = Image.fromarray(gt, mode="P")
        im.putpalette(palette)
        im.save(lab_path)
        out[name] = cls
        npix += int((gt < nc).sum())
    return out, npix


def text_tower(vocab, width=512, layers=2, embed=512, ctx=77, seed=5):
    rs = np.random.RandomState(seed)
    f = np.float32
    rn = lambda *s, std=1.0: (rs.standard_normal(s) * std).astype(f)
    w = {"token_embedding.weight": rn(vocab, width, std=0.5), "positional_embedding": rn(ctx, width, std=0.1),
         "ln_final.weight": np.ones(width, f), "ln_final.bias": np.zeros(width, f),
         "text_projection": rn(width, embed, std=width ** -0.5), "logit_scale": np.array(4.6, f)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        for nm in ("ln_1", "ln_2"):
            w[p + nm + ".weight"] = np.ones(width, f)
            w[p + nm + ".bias"] = np.zeros(width, f)
        w[p + "attn.in_proj_weight"] = rn(3 * width, width, std=width ** -0.5)
        w[p + "attn.in_proj_bias"] = np.zeros(3 * width, f)
        w[p + "attn.out_proj.weight"] = rn(width, width, std=0.5 * width ** -0.5)
        w[p + "attn.out_proj.bias"] = np.zeros(width, f)
        w[p + "mlp.c_fc.weight"] = rn(4 * width, width, std=width ** -0.5)
        w[p + "mlp.c_fc.bias"] = np.zeros(4 * width, f)
        w[p + "mlp.c_proj.weight"] = rn(width, 4 * width, std=0.5 * (4 * width) ** -0.5)
        w[p + "mlp.c_proj.bias"] = np.zeros(width, f)
    return w


def write(base, n, procs=16, seed=1234, coco=False):
    from concurrent.futures import ProcessPoolExecutor
    root, lists = os.path.join(base, "COCO" if coco else "VOC2012"), os.path.join(base, "lists")
    dirs = (os.path.join(root, "JPEGImages", "val"), os.path.join(root, "SegmentationClass", "val")) if coco else \
           (os.path.join(root, "JPEGImages"), os.path.join(root, "SegmentationClassAug"))
    for d in dirs + (lists,):
        os.makedirs(d, exist_ok=True)
    t0 = time.time()
    step = (n + 4 * procs - 1) // (4 * procs)
    jobs = [(root, lo, min(lo + step, n), n, seed, coco) for lo in range(0, n, step)]
    onehot, npix = {}, 0
    with ProcessPoolExecutor(procs) as ex:
        for o, p in ex.map(_write_slice, jobs):
            onehot.update(o)
            npix += p
    ids = sorted(onehot)
    with open(os.path.join(lists, "val.txt" if coco else "train.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    np.save(os.path.join(lists, "cls_labels_onehot.npy"), onehot)
    # checkpoint + BPE file in the published formats
    import torch
    from excel_amd.tools import synthetic
    bpe_path = os.path.join(base, "bpe_tiny_vocab.txt.gz")
    with gzip.open(bpe_path, "wb") as f:
        f.write(('"bpe_simple_vocab" - version: tiny\n' + "\n".join(MERGES)).encode("utf-8"))
    full = {"visual." + k: v for k, v in synthetic.make_vit_state_dict(seed=0).items()}
    full.update(text_tower(256 + 256 + len(MERGES) + 2))
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in full.items()}, os.path.join(base, "ViT-B-16.pt"))
    # the same parameters stored as fp16 tensors, like the published archives (clip/clip.py:138-154 loads them, clip/build_model.py:72 keeps the
    # model fp32: fp16-VALUED weights - the harness then starts in f16x2 by itself):  run ... --model <base>/ViT-B-16.fp16.pt
    os.makedirs(os.path.join(base, "fp16"), exist_ok=True)
    torch.save({k: (torch.from_numpy(np.asarray(v)).half() if np.asarray(v).dtype == np.float32 else torch.from_numpy(np.asarray(v))) for k, v in full.items()},
               os.path.join(base, "fp16", "ViT-B-16.pt"))
    nbytes = sum(os.path.getsize(os.path.join(dirs[0], i + ".jpg")) for i in ids)
    meta = {"dataset": "coco" if coco else "voc", "images": n, "scored_pixels": int(npix), "jpeg_bytes": int(nbytes), "write_seconds": round(time.time() - t0, 1)}
    json.dump(meta, open(os.path.join(base, "meta.json"), "w"))
    print("wrote", meta)


def run(base, json_out, extra):
    import logging
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    from excel_amd.tools import infer_lam
    meta = json.load(open(os.path.join(base, "meta.json")))
    coco = meta.get("dataset") == "coco"
    argv = ["--model", os.path.join(base, "ViT-B-16.pt"), "--bpe_path", os.path.join(base, "bpe_tiny_vocab.txt.gz"),
            "--list_folder", os.path.join(base, "lists"), "--json_out", json_out]
    if coco:   # BASELINE configs[4]: 80 classes + background, 512 x 512, batch 16
        argv += ["--data_folder", os.path.join(base, "COCO"), "--dataset_name", "ms_coco", "--num_classes", "81", "--infer_set", "val",
                 "--resize_size", "512", "--batch_size", "16", "--num_attri", "224"]      # (the shipped COCO bank has 224 clusters)
    else:
        argv += ["--data_folder", os.path.join(base, "VOC2012"), "--infer_set", "train", "--batch_size", "32"]
    t0 = time.time()
    score, total = infer_lam.validate(infer_lam.get_parser().parse_args(argv + extra))
    wall = time.time() - t0
    rec = json.load(open(json_out))
    tot = int(total.sum().item()) if hasattr(total, "sum") else 0
    rec.update({"workload": ("one rank's shard of BASELINE configs[4] (COCO val2014 / 8 = 5 063 images, 80 classes, 512 x 512, batch 16, single scale) on one GPU: "
                             "on-disk COCO-format tree, JPEG/PNG decode, ragged batches") if coco else
                            "BASELINE configs[3] list size on one GPU: on-disk VOC-format tree, JPEG/PNG decode, ragged batches",
                "data": "synthetic images / seeded random checkpoint in the published file formats (mIoU is plumbing)",
                "wall_seconds_incl_model_build": round(wall, 2), "scored_pixels_expected": meta["scored_pixels"], "scored_pixels": tot,
                "every_pixel_scored_once": tot == meta["scored_pixels"], "jpeg_megabytes": round(meta["jpeg_bytes"] / 1e6, 1),
                "gemm_check": getattr(infer_lam.validate, "last_gemm_check", None)})
    json.dump(rec, open(json_out, "w"), indent=1, default=str)
    print(json.dumps(rec, default=str))


if __name__ == "__main__":
    if sys.argv[1] == "write":
        write(sys.argv[2], int(sys.argv[3]), coco=len(sys.argv) > 4 and sys.argv[4] == "coco")
    else:
        run(sys.argv[2], sys.argv[3], sys.argv[4:])
