"""The oracle's two heavy stages (ViT surgery forward, PAR) restated with torch CPU fp32 kernels so that they run on ALL host
threads (numpy's element-wise softmax / exp / gather loops are single-threaded), plus the batch-1 evaluation loop of
tools/infer_lam.py:63-128 with a per-stage wall-clock split.  TEST INFRASTRUCTURE: this is bench.py's `cpu_baseline` leg (SURVEY
8d: "torch fp32, torch.set_num_threads(os.cpu_count()), batch 1, per-stage split") and is itself pinned against the numpy oracle
(tests/test_oracle_golden.py::test_torch_cpu_port_matches_numpy_oracle); nothing in excel_amd imports it.

Each function mirrors its numpy twin line by line (oracle/vit.py, oracle/par.py cite the reference lines)."""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import aff as _aff
from . import cam as _cam
from . import evaluate as _ev
from .interp import bilinear_resize
from .par import TAPS
from .vit import resize_pos_embed


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)))


class TorchVit:
    """oracle.vit.vit_forward_single on torch CPU (clip/clip_surgery_model.py:419-448, :95-159, :307)."""

    def __init__(self, w, cfg, grid):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in w.items()}
        self.pos = _t(resize_pos_embed(np.asarray(w["positional_embedding"], np.float32), grid))

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], 1e-5)

    def _qkv(self, y, p):
        h = self.cfg.heads
        qkv = F.linear(y, self.w[p + "attn.in_proj_weight"], self.w[p + "attn.in_proj_bias"])
        N, D = y.shape
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        sp = lambda t: t.reshape(N, h, D // h).permute(1, 0, 2)
        return sp(q), sp(k), sp(v)

    @staticmethod
    def _merge(t):
        h, N, d = t.shape
        return t.permute(1, 0, 2).reshape(N, h * d)

    def _mlp(self, x, p):
        hdn = F.linear(x, self.w[p + "mlp.c_fc.weight"], self.w[p + "mlp.c_fc.bias"])
        hdn = hdn * torch.sigmoid(1.702 * hdn)                                           # QuickGELU :280-282
        return F.linear(hdn, self.w[p + "mlp.c_proj.weight"], self.w[p + "mlp.c_proj.bias"])

    @torch.no_grad()
    def forward(self, img):
        """img [3,S,S] -> (x [N,out], attn_weights [L,N,N]) as numpy."""
        cfg, w = self.cfg, self.w
        L, first = cfg.layers, cfg.layers - cfg.n_surgery
        ps = cfg.patch
        x = F.conv2d(_t(img)[None], w["conv1.weight"], stride=ps)[0]                    # :421
        x = x.reshape(cfg.width, -1).t()                                                 # [P,D] :422-423
        x = torch.cat([w["class_embedding"][None], x], 0) + self.pos                     # :424-436
        x = self._ln(x, "ln_pre")
        scale = cfg.head_dim ** -0.5
        x_ori, attns = None, []
        for i in range(L):
            p = f"transformer.resblocks.{i}."
            if i < first:                                                                # nn.MultiheadAttention, head-mean weights
                q, k, v = self._qkv(self._ln(x, p + "ln_1"), p)
                a = torch.softmax((q * scale) @ k.transpose(1, 2), -1)
                o = F.linear(self._merge(a @ v), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
                x = x + o
                x = x + self._mlp(self._ln(x, p + "ln_2"), p)
                attns.append(a.mean(0))
            else:                                                                        # Attention.forward :95-159
                src = x if x_ori is None else x_ori
                q, k, v = self._qkv(self._ln(src, p + "ln_1"), p)
                attn_ori = torch.softmax((q @ k.transpose(1, 2)) * scale, -1)
                a = (torch.softmax((q @ q.transpose(1, 2)) * scale, -1) + torch.softmax((k @ k.transpose(1, 2)) * scale, -1)
                     + torch.softmax((v @ v.transpose(1, 2)) * scale, -1)) / 3
                a = a.sum(0, keepdim=True)
                Wo, bo = w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"]
                x_res = F.linear(self._merge(a @ v), Wo, bo)
                x_ori = src + F.linear(self._merge(attn_ori @ v), Wo, bo)
                x_ori = x_ori + self._mlp(self._ln(x_ori, p + "ln_2"), p)
                x = x + x_res
                attns.append(attn_ori.sum(0))
        x = x.clone()
        if x_ori is not None:
            x[0] = x_ori[0]                                                              # :442
        x = self._ln(x, "ln_post") @ w["proj"]                                           # :445-446
        return x.numpy(), torch.stack(attns[-6:], 0).numpy()                             # only the last 6 layers are consumed (affutils.py:180)


class TorchPAR:
    """oracle.par.PAR on torch CPU (utils/PAR.py:26-92); call signature of the numpy class."""

    def __init__(self, dilations, num_iter):
        self.dilations, self.num_iter = list(dilations), num_iter
        s2 = float(np.float32(np.sqrt(2)))
        ker = torch.tensor([s2, 1, s2, 1, 1, s2, 1, s2], dtype=torch.float32)
        self.pos = torch.cat([ker * float(d) for d in self.dilations])

    def _taps(self, x):
        """yield the 8*len(dil) edge-clamped shifted views of x [B,C,H,W] (F.pad(mode='replicate') + dilated one-hot conv, :39-49)
        in the reference's tap order: dilation-major, (dy,dx) = (-d,-d),(-d,0),(-d,+d),(0,-d),(0,+d),(+d,-d),(+d,0),(+d,+d)."""
        B, C, H, W = x.shape
        D = max(self.dilations)
        ys, xs = torch.arange(-D, H + D).clamp_(0, H - 1), torch.arange(-D, W + D).clamp_(0, W - 1)
        xp = x[:, :, ys][:, :, :, xs].contiguous()                                       # replicate padding by D, once
        for d in self.dilations:
            for ty, tx in TAPS:
                yield xp[:, :, D + ty * d:D + ty * d + H, D + tx * d:D + tx * d + W]

    @torch.no_grad()
    def __call__(self, imgs, masks):
        masks = _t(masks)
        B, C, H, W = masks.shape
        im = _t(bilinear_resize(np.asarray(imgs, np.float32), H, W, align_corners=True))       # :67
        nb = torch.stack(list(self._taps(im)), 2)                                        # [B,3,48,H,W]
        absd = (nb - im[:, :, None]).abs()
        std = nb.std(dim=2, keepdim=True)                                                # unbiased :77
        aff = -((absd / (std + 1e-8) / 0.3) ** 2)
        aff = aff.mean(1, keepdim=True)
        pos_aff = -((self.pos / (self.pos.std() + 1e-8) / 0.3) ** 2)
        aff = (torch.softmax(aff, 2) + 0.01 * torch.softmax(pos_aff, 0).reshape(1, 1, -1, 1, 1))[:, 0]    # [B,48,H,W]
        for _ in range(self.num_iter):                                                   # :88-90, accumulated tap by tap (same order)
            out = torch.zeros_like(masks)
            for t, v in enumerate(self._taps(masks)):
                out.addcmul_(v, aff[:, t:t + 1])
            masks = out
        return masks.numpy()


def build_validation(samples, w, cfg, text_attr, num_classes=21, resize_size=448, dilations=(1, 2, 4, 8, 12, 24), num_iter=20,
                     caa_thre=0.79, threads=None, time_budget_s=None, min_images=1):
    """tools/infer_lam.py:63-128 at batch 1 (:167) -> (hist, labels, seconds per stage).  Heavy stages on torch CPU fp32 with
    `threads` host threads (default: all), the small ones (CAM epilogue, Sinkhorn, boxes, resize) on the numpy oracle.
    `samples` may be a generator; with `time_budget_s` the loop stops after the first image that exceeds it (>= min_images)."""
    if threads:
        torch.set_num_threads(int(threads))
    vit = TorchVit(w, cfg, resize_size // cfg.patch)
    par = TorchPAR(dilations, num_iter)
    stage = dict(vit=0.0, cam=0.0, aff_random_walk=0.0, upsample_par_argmax=0.0, score=0.0)
    gts, preds, per_image = [], [], []
    t_start = time.perf_counter()
    for img, gt, cls_label in samples:
        t0 = time.perf_counter()
        inputs = bilinear_resize(np.asarray(img, np.float32)[None], resize_size, resize_size, align_corners=False)   # :74
        x, attn = vit.forward(inputs[0])
        t1 = time.perf_counter()
        f = x[None] / np.sqrt((x[None] * x[None]).sum(axis=1, keepdims=True, dtype=np.float32))                     # clip.py:353
        maps = _cam.clip_feature_surgery(f.astype(np.float32), text_attr.T)[:, 1:, :num_classes - 1]                 # :79
        t2 = time.perf_counter()
        refined, cls_lst = _aff.refine_cams_with_aff(maps[0], attn, cls_label, size=inputs.shape[2:], caa_thre=caa_thre)   # :93
        t3 = time.perf_counter()
        label, _ = _aff.refine_cams_with_bkg_weclip(refined, inputs[0], cls_lst, par, gt.shape[-2:])                # :94
        t4 = time.perf_counter()
        preds.append(label[0].astype(np.int16))
        gts.append(np.asarray(gt).astype(np.int16))
        for k, d in zip(("vit", "cam", "aff_random_walk", "upsample_par_argmax"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            stage[k] += d
        per_image.append(t4 - t0)
        if time_budget_s is not None and len(preds) >= min_images and time.perf_counter() - t_start > time_budget_s:
            break
    t0 = time.perf_counter()
    hist = _ev.hist_of(gts, preds, num_classes)                                                                     # :121
    stage["score"] = time.perf_counter() - t0
    build_validation.last_per_image_seconds = per_image        # (per-image wall times of the last call: bench.py reports their quartiles)
    return hist, preds, stage
