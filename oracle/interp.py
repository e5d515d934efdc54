"""Bilinear resampling restated in numpy (oracle; test infrastructure only).

torch.nn.functional.interpolate(mode='bilinear') is a third-party (PyTorch)
op used by the reference at
  * tools/infer_lam.py:74          (input resize, align_corners=False)
  * clip/clip_surgery_model.py:410,432 (positional-embedding resize, default
    align_corners=False)
  * utils/PAR.py:67                (guide image resize, align_corners=True)
Its published sampling rule (ATen UpSample.h, area_pixel_compute_source_index)
is restated here; coefficient arithmetic is done in float32 like ATen's CPU
kernel for float tensors.
"""
import numpy as np


def _src_index(out_size, in_size, align_corners):
    """Returns (i0, i1, lam) as (int64, int64, float32) arrays of length out_size."""
    dst = np.arange(out_size, dtype=np.float32)
    if align_corners:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
        src = scale * dst
    else:
        scale = np.float32(in_size) / np.float32(out_size)
        src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
        src = np.maximum(src, np.float32(0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def bilinear_resize(x, out_h, out_w, align_corners=False):
    """x: [..., H, W] float32 -> [..., out_h, out_w] float32."""
    x = np.asarray(x, dtype=np.float32)
    H, W = x.shape[-2:]
    if (H, W) == (out_h, out_w):
        # scale 1: every mode samples the source pixel exactly
        return x.copy()
    y0, y1, ly = _src_index(out_h, H, align_corners)
    x0, x1, lx = _src_index(out_w, W, align_corners)
    hy = (np.float32(1) - ly)[:, None]
    ly = ly[:, None]
    hx = (np.float32(1) - lx)[None, :]
    lx = lx[None, :]
    r0 = x[..., y0, :]
    r1 = x[..., y1, :]
    # ATen: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
    top = hx * r0[..., :, x0] + lx * r0[..., :, x1]
    bot = hx * r1[..., :, x0] + lx * r1[..., :, x1]
    return (hy * top + ly * bot).astype(np.float32)


def cv2_resize_linear(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h)) with the default INTER_LINEAR for a 2-D
    float32 image (used at utils/affutils.py:75).  OpenCV is third-party and
    absent here (PARITY UNPINNED); this restates its documented rule:
    half-pixel centres, source index clamped to the edge, coordinate in double
    then weight in float, horizontal pass then vertical pass.
    """
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape

    def coef(dst_n, src_n):
        scale = 1.0 / (float(dst_n) / float(src_n))
        d = np.arange(dst_n, dtype=np.float64)
        f = (d + 0.5) * scale - 0.5
        s = np.floor(f).astype(np.int64)
        f = (f - s).astype(np.float32)
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= src_n - 1
        f[hi] = 0
        s[hi] = src_n - 1
        s1 = np.minimum(s + 1, src_n - 1)
        return s, s1, f

    x0, x1, fx = coef(out_w, W)
    y0, y1, fy = coef(out_h, H)
    rows = img[:, x0] * (np.float32(1) - fx)[None, :] + img[:, x1] * fx[None, :]
    out = rows[y0, :] * (np.float32(1) - fy)[:, None] + rows[y1, :] * fy[:, None]
    return out.astype(np.float32)
