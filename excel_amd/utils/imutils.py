"""Host-side format helpers either side of the path: mirror of the live parts of utils/imutils.py and of the on-disk
records tools/infer_lam.py exchanges with its CRF stage.

  colormap / encode_cmap   utils/imutils.py:7-9, :32-50   (the PASCAL VOC palette: bit-interleaved class index)
  save_logits/load_logits  tools/infer_lam.py:116-119 (writer), :203-206 (reader): np.save of the dict
                           {"valid_lam": cams [k+1,H,W] f32, "keys_gt": present classes int64}
  crf_keys_to_labels       tools/infer_lam.py:225-227: keys = pad(keys_gt + 1, (1, 0)); label = keys[argmax]
  save_label_png           the colour-coded label image the reference writes with imageio (:228); PIL here

Plain numpy / PIL on the host: these are file formats, not compute.  DenseCRF itself (utils/dcrf.py) is excel_amd/utils/dcrf.py over
excel_dcrf_inference (crf.hip).
"""
import os

import numpy as np


def colormap(N=256, normalized=False):
    """VOC palette: colour channel bits are the class index's bits taken 3 at a time, MSB first (imutils.py:32-50)."""
    idx = np.arange(N, dtype=np.int64)
    cmap = np.zeros((N, 3), dtype=np.int64)
    c = idx.copy()
    for j in range(8):
        for ch in range(3):
            cmap[:, ch] |= ((c >> ch) & 1) << (7 - j)
        c >>= 3
    return (cmap / 255).astype(np.float32) if normalized else cmap.astype(np.uint8)


def encode_cmap(label):
    """label [H,W] (any integer dtype; 255 = ignore -> white-ish palette entry) -> RGB uint8 [H,W,3] (imutils.py:7-9)."""
    return colormap()[np.asarray(label).astype(np.int16), :]


def denormalize_img(imgs=None, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
    """utils/imutils.py:11-19 (device tensor in, uint8 device tensor out; HIP kernel)."""
    from .. import ops
    return ops.denormalize_img(imgs, mean, std, as_float=False)


def denormalize_img2(imgs=None):
    """utils/imutils.py:21-25: denormalize_img(imgs) / 255.0 (the PAR guide image of the training loop, scripts/train_voc.py:181)."""
    from .. import ops
    return ops.denormalize_img(imgs, as_float=True)


def save_logits(logits_dir, name, valid_lam, keys_gt, run_token=None):
    """tools/infer_lam.py:116-119.  `run_token` (optional, an extra key next to the reference's two: its reader looks keys up by name)
    marks the run that wrote the record, so the CRF stage can refuse records of an earlier run whatever the file system's time
    stamps say."""
    os.makedirs(logits_dir, exist_ok=True)
    valid_lam = valid_lam.detach().cpu().numpy() if hasattr(valid_lam, "detach") else np.asarray(valid_lam)
    keys_gt = keys_gt.detach().cpu().numpy() if hasattr(keys_gt, "detach") else np.asarray(keys_gt)
    path = os.path.join(logits_dir, name + ".npy")
    rec = {"valid_lam": valid_lam, "keys_gt": keys_gt}
    if run_token is not None:
        rec["run_token"] = str(run_token)
    np.save(path, rec)
    return path


def logits_run_token(path):
    """The run token of a record written by save_logits (None: written without one, e.g. by the reference)."""
    return np.load(path, allow_pickle=True).item().get("run_token")


def load_logits(path):
    """tools/infer_lam.py:203-206 -> (valid_lam, keys_gt)."""
    d = np.load(path, allow_pickle=True).item()
    return d["valid_lam"], d["keys_gt"]


def crf_keys_to_labels(prob, keys_gt):
    """tools/infer_lam.py:225-227: prob [k+1,H,W] -> uint8 labels through keys = [0, keys_gt + 1...]."""
    pred = np.argmax(prob, axis=0)
    keys = np.pad(np.asarray(keys_gt) + 1, (1, 0), mode="constant")
    return keys[pred].astype(np.uint8)


def save_label_png(path, label):
    """Colour-coded label image (tools/infer_lam.py:228)."""
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(encode_cmap(np.squeeze(label)).astype(np.uint8)).save(path)
    return path
