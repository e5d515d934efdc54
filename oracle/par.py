"""Pixel-adaptive refinement (PAR) restated in numpy fp32 (oracle; test infrastructure only).

Follows utils/PAR.py of the reference:
  get_kernel :10-24, get_dilated_neighbors :39-49, get_pos :51-62, forward :64-92.
F.pad(mode='replicate') + a dilated 3x3 conv with one-hot taps is a gather with
edge-clamped indices; tap order per dilation is (dy,dx) =
(-d,-d),(-d,0),(-d,+d),(0,-d),(0,+d),(+d,-d),(+d,0),(+d,+d), dilation-major.
"""
import numpy as np

from .interp import bilinear_resize
from .vit import softmax

TAPS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


class PAR:
    def __init__(self, dilations, num_iter):
        self.dilations = list(dilations)
        self.num_iter = num_iter
        self.w1 = np.float32(0.3)
        self.w2 = np.float32(0.01)
        s2 = np.float32(np.sqrt(2))
        ker = np.array([s2, 1, s2, 1, 1, s2, 1, s2], np.float32)                       # :54-58
        self.pos = np.concatenate([ker * np.float32(d) for d in self.dilations])       # :60-62

    def neighbors(self, x):
        """x [B,C,H,W] -> [B,C,8*len(dil),H,W]"""
        B, C, H, W = x.shape
        ys = np.arange(H)
        xs = np.arange(W)
        out = []
        for d in self.dilations:
            for (ty, tx) in TAPS:
                yy = np.clip(ys + ty * d, 0, H - 1)
                xx = np.clip(xs + tx * d, 0, W - 1)
                out.append(x[:, :, yy][:, :, :, xx])
        return np.stack(out, 2)

    def affinity(self, imgs, H, W):
        imgs = bilinear_resize(np.asarray(imgs, np.float32), H, W, align_corners=True)  # :67
        nb = self.neighbors(imgs)                                                       # :70
        absd = np.abs(nb - imgs[:, :, None])                                            # :76
        std = nb.std(axis=2, ddof=1, keepdims=True, dtype=np.float32)                   # :77 (unbiased)
        aff = -((absd / (std + np.float32(1e-8)) / self.w1) ** 2)                       # :80
        aff = aff.mean(1, keepdims=True, dtype=np.float32)                              # :81
        pos_std = self.pos.std(ddof=1, dtype=np.float32)                                # :78
        pos_aff = -((self.pos / (pos_std + np.float32(1e-8)) / self.w1) ** 2)           # :83
        aff = softmax(aff, 2) + self.w2 * softmax(pos_aff, 0).reshape(1, 1, -1, 1, 1)   # :86
        return aff.astype(np.float32)

    def __call__(self, imgs, masks):
        masks = np.asarray(masks, np.float32)
        B, C, H, W = masks.shape
        aff = self.affinity(imgs, H, W)
        for _ in range(self.num_iter):                                                  # :88-90
            masks = (self.neighbors(masks) * aff).sum(2, dtype=np.float32)
        return masks.astype(np.float32)
