"""torch.Tensor-facing wrappers of the C ABI (device memory + stream plumbing only).

Every function here hands raw device pointers of CUDA(=HIP) tensors to
libexcel_hip; nothing is computed with torch ops.  Tensors must be fp32,
contiguous and on the GPU; outputs are fresh tensors.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, lib

PAR_DILATIONS = (1, 2, 4, 8, 12, 24)


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("excel_amd ops need GPU tensors (the HIP library is the only compute path)")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def f32c(t):
    return t.to(dtype=torch.float32).contiguous()


# ------------------------------------------------------------------ building blocks
def gemm(A, Bm, bias=None, residual=None, act=0, b_kmajor=True):
    """C = act(A @ op(B) + bias) + residual.  A [M,K] (or [batch,M,K]); B [N,K] if b_kmajor else [K,N]."""
    batched = A.dim() == 3
    A3 = A if batched else A[None]
    B3 = Bm if Bm.dim() == 3 else Bm[None]
    batch, M, K = A3.shape
    N = B3.shape[1] if b_kmajor else B3.shape[2]
    out = torch.empty((batch, M, N), dtype=torch.float32, device=A.device)
    sB = 0 if B3.shape[0] == 1 else B3.stride(0)
    check(lib().excel_gemm_f32(_p(A3), _p(B3), _p(out), _p(bias), _p(residual), M, N, K, A3.stride(1), B3.stride(1), N,
                               N, 1 if b_kmajor else 0, act, batch, A3.stride(0), sB, M * N, M * N, _stream()), "excel_gemm_f32")
    return out if batched else out[0]


GEMM_MODES = {"f32": 0, "bf16x3": 1, "f16x3": 2, "f16x2": 3}      # excel_vit_set_gemm_mode (include/excel_hip.h)


def split_bf16(x, f16=False):
    """fp32 [R,K] -> split operand [R,2,K] (bf16 - or, with f16, IEEE-half - bit patterns in an int16 tensor)."""
    x = f32c(x)
    R, K = x.shape
    out = torch.empty((R, 2, K), dtype=torch.int16, device=x.device)
    fn = lib().excel_split_f16 if f16 else lib().excel_split_bf16
    check(fn(_p(x), _p(out, torch.int16), R, K, _stream()), "excel_split_f16" if f16 else "excel_split_bf16")
    return out


def gemm_bf16x3(A_split, W_split, bias=None, residual=None, act=0, split_out=False, f16=False):
    """C = act(A . W^T + bias) + residual from two split operands (three 16-bit MFMAs per product); f16: IEEE-half planes (split_bf16(f16=True))."""
    M, _, K = A_split.shape
    N = W_split.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=A_split.device)
    fn = lib().excel_gemm_f16x3 if f16 else lib().excel_gemm_bf16x3
    check(fn(_p(A_split, torch.int16), _p(W_split, torch.int16), _p(out), _p(bias), _p(residual), M, N, K, act,
             1 if split_out else 0, _stream()), "excel_gemm_f16x3" if f16 else "excel_gemm_bf16x3")
    return out.view(torch.int16).view(M, 2, N) if split_out else out


def pack_f16(x):
    """fp32 [R,K] -> (the hi plane as a plain IEEE-half matrix [R,K] (int16 bit patterns), number of elements NOT representable in
    half).  The second value is 0 exactly when `x` is fp16-valued - the precondition of gemm_f16x2 (every published CLIP archive)."""
    x = f32c(x)
    R, K = x.shape
    out = torch.empty((R, K), dtype=torch.int16, device=x.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=x.device)
    check(lib().excel_pack_f16(_p(x), _p(out, torch.int16), R, K, _p(cnt, torch.int64), _stream()), "excel_pack_f16")
    return out, int(cnt.item())


def gemm_f16x2(A_split, W_split, W_half=None, bias=None, residual=None, act=0, split_out=False):
    """gemm_bf16x3(..., f16=True) for fp16-VALUED weights (the lo plane of W_split is all zero): two MFMAs per product, the same bits.
    W_half (pack_f16 of the same weights) lets the large-tile kernel stream half the weight bytes."""
    M, _, K = A_split.shape
    N = W_split.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=A_split.device)
    check(lib().excel_gemm_f16x2(_p(A_split, torch.int16), _p(W_split, torch.int16), _p(W_half, torch.int16) if W_half is not None else None,
                                 _p(out), _p(bias), _p(residual), M, N, K, act, 1 if split_out else 0, _stream()), "excel_gemm_f16x2")
    return out.view(torch.int16).view(M, 2, N) if split_out else out


def layernorm(x, w, b, eps=1e-5):
    D = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().excel_layernorm(_p(x), _p(w), _p(b), _p(y), x.numel() // D, D, eps, _stream()), "excel_layernorm")
    return y


# ------------------------------------------------------------------ ViT
class VitHandle:
    """Owns device copies of the ViT weights (state_dict naming of the reference's VisionTransformer,
    clip/clip_surgery_model.py:374-394) and the C handle bound to them."""

    def __init__(self, state_dict, width, layers, heads, patch, out_dim, n_surgery=5, device="cuda", prefix="", gemm_mode=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("excel_amd.VitHandle needs a GPU device: the HIP library is the only compute path")
        g = lambda k: f32c(torch.as_tensor(state_dict[prefix + k])).to(self.device)
        self.cfg = dict(width=width, layers=layers, heads=heads, patch=patch, out_dim=out_dim, n_surgery=n_surgery)
        self.t = {}
        top = {"conv1_w": "conv1.weight", "class_emb": "class_embedding", "pos_emb": "positional_embedding",
               "ln_pre_w": "ln_pre.weight", "ln_pre_b": "ln_pre.bias", "ln_post_w": "ln_post.weight",
               "ln_post_b": "ln_post.bias", "proj": "proj"}
        for f, k in top.items():
            self.t[f] = g(k)
        pos_rows = self.t["pos_emb"].shape[0]
        pos_grid = int(round((pos_rows - 1) ** 0.5))
        assert pos_grid * pos_grid + 1 == pos_rows
        self.blocks = (_lib.VitBlockWeights * layers)()
        for i in range(layers):
            p = f"transformer.resblocks.{i}."
            # surgery blocks of a reloaded reference model carry attn.qkv/attn.proj instead of in_proj/out_proj (:401-404)
            def pick(*names):
                for n in names:
                    if prefix + p + n in state_dict:
                        return g(p + n)
                raise KeyError(p + names[0])
            fields = {
                "ln1_w": pick("ln_1.weight"), "ln1_b": pick("ln_1.bias"),
                "in_proj_w": pick("attn.in_proj_weight", "attn.qkv.weight"),
                "in_proj_b": pick("attn.in_proj_bias", "attn.qkv.bias"),
                "out_proj_w": pick("attn.out_proj.weight", "attn.proj.weight"),
                "out_proj_b": pick("attn.out_proj.bias", "attn.proj.bias"),
                "ln2_w": pick("ln_2.weight"), "ln2_b": pick("ln_2.bias"),
                "fc1_w": pick("mlp.c_fc.weight"), "fc1_b": pick("mlp.c_fc.bias"),
                "fc2_w": pick("mlp.c_proj.weight"), "fc2_b": pick("mlp.c_proj.bias"),
            }
            for f, t in fields.items():
                self.t[f"{i}.{f}"] = t
                setattr(self.blocks[i], f, t.data_ptr())
        w = _lib.VitWeights()
        for f in top:
            setattr(w, f, self.t[f].data_ptr())
        w.blocks = self.blocks
        cfg = _lib.VitConfig(width, layers, heads, patch, out_dim, n_surgery, pos_grid)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().excel_vit_create(C.byref(cfg), C.byref(w), C.byref(self._h)), "excel_vit_create")
            # numerics of the linear layers and attention scores: "bf16x3" (default; fp32 operands as bf16 hi+lo, 3 bf16
            # MFMAs per product, CAM error ~1e-5 against exact fp32 on well-conditioned weights), "f16x3" (IEEE-half planes: fp32-grade
            # results, ~2.5 % slower), "f16x2" (f16x3 with two MFMAs per product in the nn.Linear GEMMs - needs fp16-VALUED weights, which
            # every published CLIP archive has: bit-identical to f16x3 there and faster than bf16x3) or "f32" (exact fp32 MFMA).
            # Default ("auto"): f16x2 when the weights allow it, else bf16x3.  EXCEL_GEMM_MODE overrides.
            mode = gemm_mode or os.environ.get("EXCEL_GEMM_MODE", "auto")
            if mode != "f32" and (width % 32 or (3 * patch * patch) % 32):
                mode = "f32"
            if mode == "auto":
                mode = "f16x2" if self.weights_fp16_exact() else "bf16x3"
            self.set_gemm_mode(mode)
        self._ws = {}          # one workspace per launch stream: the same weights can serve concurrent streams

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().excel_vit_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_gemm_mode(self, mode):
        """'f32' (exact fp32 MFMA), 'bf16x3' (split-bf16 operands, 3 bf16 MFMAs per product), 'f16x3' (IEEE-half planes) or 'f16x2'
        (f16x3 with two-product weight GEMMs; raises unless weights_fp16_exact())."""
        check(lib().excel_vit_set_gemm_mode(self._h, GEMM_MODES[mode]), "excel_vit_set_gemm_mode")

    def weights_fp16_exact(self):
        """True when every GEMM weight of this handle is exactly representable in IEEE half (clip/build_model.py:72 loads the published
        fp16 archive into the fp32 model unchanged, so real CLIP weights are)."""
        rc = lib().excel_vit_weights_fp16_exact(self._h)
        if rc < 0:
            check(rc, "excel_vit_weights_fp16_exact")
        return rc == 1

    def gemm_mode(self):
        return {v: k for k, v in GEMM_MODES.items()}[lib().excel_vit_get_gemm_mode(self._h)]

    def workspace(self, B, S):
        need = lib().excel_vit_workspace_bytes(self._h, B, S)
        if need == 0:
            raise ValueError(f"bad ViT input shape B={B} S={S}")
        key = torch.cuda.current_stream().cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)
            ws = None
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws, need

    def forward(self, imgs, want_w_aff=True, aff_layers=6, n_attn_out=0, want_feats=False, want_raw=False, ex_attn=None,
                feats_as_reference=False, want_features=True):
        """-> dict(image_features [B,N,C], w_aff [B,P,P]|None, attn [n,B,N,N]|None, feats [L,B,N,D]|None, x_raw|None)
        ex_attn [B,P,P]: LVC cue added to every head of every surgery block (clip_surgery_model.py:127-141).
        feats_as_reference: `feats` as the reference's decoder receives them (in-place aliasing quirk, include/excel_hip.h)."""
        imgs = f32c(imgs)
        B, _, S, S2 = imgs.shape
        assert S == S2
        c = self.cfg
        g = S // c["patch"]
        N = g * g + 1
        dev = imgs.device
        if not (want_features or want_raw):
            raise ValueError("VitHandle.forward: want_features or want_raw")
        # want_features=False: only x_raw (the fused CAM kernel normalises over the token axis itself)
        f = torch.empty((B, N, c["out_dim"]), dtype=torch.float32, device=dev) if want_features else None
        raw = torch.empty((B, N, c["out_dim"]), dtype=torch.float32, device=dev) if want_raw else None
        w_aff = torch.empty((B, N - 1, N - 1), dtype=torch.float32, device=dev) if want_w_aff else None
        attn = torch.empty((n_attn_out, B, N, N), dtype=torch.float32, device=dev) if n_attn_out else None
        feats = torch.empty((c["layers"], B, N, c["width"]), dtype=torch.float32, device=dev) if want_feats else None
        ws, need = self.workspace(B, S)
        if ex_attn is not None:
            ex_attn = f32c(ex_attn)
            if tuple(ex_attn.shape) != (B, N - 1, N - 1):
                raise ValueError(f"ex_attn must be [B,P,P] = {(B, N - 1, N - 1)}, got {tuple(ex_attn.shape)}")
        check(lib().excel_vit_forward_ex(self._h, _p(imgs), B, S, _p(ws, torch.uint8), need, _p(f), _p(raw), _p(w_aff),
                                         aff_layers, _p(attn), n_attn_out, _p(feats), _p(ex_attn), 1 if feats_as_reference else 0, _stream()),
              "excel_vit_forward")
        return dict(image_features=f, w_aff=w_aff, attn=attn, feats=feats, x_raw=raw)


class DecoderHandle:
    """Device copies of the decoder head's weights + the C handle (SegFormerHead fuse: model/segformer_head.py:47-77;
    DecoderTransformer: model/decoder/TransDecoder.py:105-124).  `fuse_sd` / `dec_sd` use the reference modules' state_dict keys."""

    def __init__(self, fuse_sd, dec_sd, heads=8, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("excel_amd.DecoderHandle needs a GPU device: the HIP library is the only compute path")
        g = lambda sd, k: f32c(torch.as_tensor(sd[k])).to(self.device)
        self.t = {}
        L = 0
        while f"linears_modulelist.{L}.proj.weight" in fuse_sd:
            L += 1
        nl = 0
        while f"transformer.resblocks.{nl}.ln_1.weight" in dec_sd:
            nl += 1
        E, D = fuse_sd["linears_modulelist.0.proj.weight"].shape
        nc = dec_sd["linear_pred.weight"].shape[0]
        self.cfg = dict(vit_layers=L, vit_width=int(D), embed=int(E), dec_layers=nl, heads=heads, num_classes=int(nc))
        self.fuse = (_lib.FuseLayerWeights * L)()
        for l in range(L):
            for f, k in (("proj_w", "proj.weight"), ("proj_b", "proj.bias"), ("proj2_w", "proj_2.weight"), ("proj2_b", "proj_2.bias")):
                t = self.t[f"fuse{l}.{f}"] = g(fuse_sd, f"linears_modulelist.{l}.{k}")
                setattr(self.fuse[l], f, t.data_ptr())
        self.blocks = (_lib.DecoderBlockWeights * max(nl, 1))()
        names = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "in_proj_w": "attn.in_proj_weight", "in_proj_b": "attn.in_proj_bias",
                 "out_proj_w": "attn.out_proj.weight", "out_proj_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
                 "fc1_w": "mlp.c_fc.weight", "fc1_b": "mlp.c_fc.bias", "fc2_w": "mlp.c_proj.weight", "fc2_b": "mlp.c_proj.bias"}
        for l in range(nl):
            for f, k in names.items():
                t = self.t[f"blk{l}.{f}"] = g(dec_sd, f"transformer.resblocks.{l}.{k}")
                setattr(self.blocks[l], f, t.data_ptr())
        w = _lib.DecoderWeights()
        w.fuse, w.blocks = self.fuse, self.blocks
        for f, (sd, k) in {"fuse_w": (fuse_sd, "linear_fuse.weight"), "fuse_b": (fuse_sd, "linear_fuse.bias"),
                           "pred_w": (dec_sd, "linear_pred.weight"), "pred_b": (dec_sd, "linear_pred.bias")}.items():
            t = self.t[f] = g(sd, k).reshape(sd[k].shape[0], -1).contiguous() if f.endswith("_w") else g(sd, k)
            setattr(w, f, t.data_ptr())
        cfg = _lib.DecoderConfig(L, int(D), int(E), nl, heads, int(nc))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().excel_decoder_create(C.byref(cfg), C.byref(w), C.byref(self._h)), "excel_decoder_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().excel_decoder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def state_dicts(self):
        """-> (fuse_sd, dec_sd) with the reference modules' state_dict keys and shapes (1x1 conv weights as [out,in,1,1]): what
        torch.save(model.state_dict()) of the reference holds for the head, so a head trained here loads back there."""
        c = self.cfg
        fuse, dec = {}, {}
        for l in range(c["vit_layers"]):
            for f, k in (("proj_w", "proj.weight"), ("proj_b", "proj.bias"), ("proj2_w", "proj_2.weight"), ("proj2_b", "proj_2.bias")):
                fuse[f"linears_modulelist.{l}.{k}"] = self.t[f"fuse{l}.{f}"].detach().clone()
        fuse["linear_fuse.weight"] = self.t["fuse_w"].detach().clone()[:, :, None, None]
        fuse["linear_fuse.bias"] = self.t["fuse_b"].detach().clone()
        for l in range(c["dec_layers"]):
            for f, k in _BLOCK_KEYS.items():
                dec[f"transformer.resblocks.{l}.{k}"] = self.t[f"blk{l}.{f}"].detach().clone()
        dec["linear_pred.weight"] = self.t["pred_w"].detach().clone()[:, :, None, None]
        dec["linear_pred.bias"] = self.t["pred_b"].detach().clone()
        return fuse, dec

    # ------------------------------------------------------------------ training iteration
    def _grad_table(self):
        """Gradient tensors shaped like the parameters + the C table pointing at them (built once)."""
        if getattr(self, "_grads", None) is None:
            # one flat buffer (each parameter's gradient is a 16-byte aligned view): a data-parallel step is ONE all-reduce
            offs, total = {}, 0
            for k, v in self.t.items():
                offs[k] = total
                total += (v.numel() + 3) // 4 * 4
            self.grad_flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            self._grads = {k: self.grad_flat[offs[k]:offs[k] + v.numel()].view(v.shape) for k, v in self.t.items()}
            c = self.cfg
            self._gfuse = (_lib.FuseLayerWeights * c["vit_layers"])()
            for l in range(c["vit_layers"]):
                for f in ("proj_w", "proj_b", "proj2_w", "proj2_b"):
                    setattr(self._gfuse[l], f, self._grads[f"fuse{l}.{f}"].data_ptr())
            self._gblocks = (_lib.DecoderBlockWeights * max(c["dec_layers"], 1))()
            for l in range(c["dec_layers"]):
                for f in _BLOCK_KEYS:
                    setattr(self._gblocks[l], f, self._grads[f"blk{l}.{f}"].data_ptr())
            g = _lib.DecoderWeights()
            g.fuse, g.blocks = self._gfuse, self._gblocks
            for f in ("fuse_w", "fuse_b", "pred_w", "pred_b"):
                setattr(g, f, self._grads[f].data_ptr())
            self._gtable = g
        return self._grads, self._gtable

    def _check_feats(self, L, N, D, g):
        """all_feats [L,B,N,D] must match the head's configuration: the kernels index by it (a wrong shape would read out of bounds)."""
        c = self.cfg
        if L != c["vit_layers"] or D != c["vit_width"]:
            raise ValueError(f"all_feats must be [{c['vit_layers']},B,N,{c['vit_width']}], got L={L}, D={D}")
        if g * g + 1 != N:
            raise ValueError("all_feats must hold cls + a square grid of patch tokens")

    def forward_train(self, all_feats, dropout_p=0.0, dropout_seed=0):
        """-> (seg [B,nc,g,g], attn_pred [B,P,P], ctx) ; ctx goes to backward().  dropout_p: the head's Dropout2d (0.1 in the reference)."""
        all_feats = f32c(all_feats)
        L, B, N, D = all_feats.shape
        g = int(round((N - 1) ** 0.5))
        c = self.cfg
        self._check_feats(L, N, D, g)
        dev = all_feats.device
        seg = torch.empty((B, c["num_classes"], g, g), dtype=torch.float32, device=dev)
        ap = torch.empty((B, g * g, g * g), dtype=torch.float32, device=dev)
        need = lib().excel_decoder_train_workspace_bytes(self._h, B, g)
        ws = _ws(need, dev)
        check(lib().excel_decoder_forward_train(self._h, _p(all_feats), B, g, _p(ws, torch.uint8), need, _p(seg), _p(ap), float(dropout_p),
                                                int(dropout_seed) & 0xFFFFFFFF, _stream()), "excel_decoder_forward_train")
        return seg, ap, (all_feats, B, g, ws, need, float(dropout_p), int(dropout_seed) & 0xFFFFFFFF)

    def train_attn_fts(self, ctx):
        """attn_fts [B,E,g,g] of the forward_train that produced `ctx`: the post-Dropout2d fused features (the LVC cue of the
        training loop, scripts/train_voc.py:186-189)."""
        all_feats, B, g, ws, need, _, _ = ctx
        fts = torch.empty((B, self.cfg["embed"], g, g), dtype=torch.float32, device=all_feats.device)
        check(lib().excel_decoder_train_attn_fts(self._h, B, g, _p(ws, torch.uint8), need, _p(fts), _stream()), "excel_decoder_train_attn_fts")
        return fts

    def backward(self, ctx, d_seg, d_attn_pred=None):
        """-> dict of gradients keyed like self.t (fuse{l}.proj_w ..., blk{l}.fc1_w ..., fuse_w, pred_w ...)."""
        all_feats, B, g, ws, need, dp, dseed = ctx
        grads, table = self._grad_table()
        check(lib().excel_decoder_backward(self._h, _p(all_feats), B, g, _p(ws, torch.uint8), need, _p(f32c(d_seg)),
                                           _p(f32c(d_attn_pred)) if d_attn_pred is not None else None, C.byref(table), dp, dseed, _stream()),
              "excel_decoder_backward")
        return grads

    def adamw_step(self, lr, step, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        """torch.optim.AdamW over every parameter with the gradients of the last backward(); `step` counts from 1."""
        grads, _ = self._grad_table()
        if getattr(self, "_adam", None) is None:
            self._adam = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.t.items()}
        for k, p in self.t.items():
            m, v = self._adam[k]
            check(lib().excel_adamw_step(_p(p), _p(grads[k]), _p(m), _p(v), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                         float(weight_decay), int(step), _stream()), "excel_adamw_step")

    def forward(self, all_feats, want_seg=True):
        """all_feats [L,B,N,D] -> (attn_fts [B,E,g,g], seg [B,nc,g,g] | None)"""
        all_feats = f32c(all_feats)
        L, B, N, D = all_feats.shape
        c = self.cfg
        g = int(round((N - 1) ** 0.5))
        self._check_feats(L, N, D, g)
        dev = all_feats.device
        fts = torch.empty((B, c["embed"], g, g), dtype=torch.float32, device=dev)
        seg = torch.empty((B, c["num_classes"], g, g), dtype=torch.float32, device=dev) if want_seg else None
        need = lib().excel_decoder_workspace_bytes(self._h, B, g)
        ws = _ws(need, dev)
        check(lib().excel_decoder_forward(self._h, _p(all_feats), B, g, _p(ws, torch.uint8), need, _p(fts), _p(seg), _stream()),
              "excel_decoder_forward")
        return fts, seg


_BLOCK_KEYS = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "in_proj_w": "attn.in_proj_weight", "in_proj_b": "attn.in_proj_bias",
               "out_proj_w": "attn.out_proj.weight", "out_proj_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
               "fc1_w": "mlp.c_fc.weight", "fc1_b": "mlp.c_fc.bias", "fc2_w": "mlp.c_proj.weight", "fc2_b": "mlp.c_proj.bias"}


class TextHandle:
    """CLIP text tower (encode_text, clip/clip_surgery_model.py:551-564) over the HIP library.  `sd`: the CLIP state_dict keys of
    the text side (token_embedding.weight, positional_embedding, transformer.resblocks.*, ln_final.*, text_projection)."""

    def __init__(self, sd, heads, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("excel_amd.TextHandle needs a GPU device: the HIP library is the only compute path")
        g = lambda k: f32c(torch.as_tensor(sd[k])).to(self.device)
        self.t = {}
        nl = 0
        while f"transformer.resblocks.{nl}.ln_1.weight" in sd:
            nl += 1
        self.blocks = (_lib.DecoderBlockWeights * max(nl, 1))()
        for l in range(nl):
            for f, k in _BLOCK_KEYS.items():
                t = self.t[f"blk{l}.{f}"] = g(f"transformer.resblocks.{l}.{k}")
                setattr(self.blocks[l], f, t.data_ptr())
        w = _lib.TextWeights()
        for f, k in (("token_embedding", "token_embedding.weight"), ("positional_embedding", "positional_embedding"),
                     ("ln_final_w", "ln_final.weight"), ("ln_final_b", "ln_final.bias"), ("text_projection", "text_projection")):
            t = self.t[f] = g(k)
            setattr(w, f, t.data_ptr())
        w.blocks = self.blocks
        vocab, width = self.t["token_embedding"].shape
        ctx = self.t["positional_embedding"].shape[0]
        embed = self.t["text_projection"].shape[1]
        self.cfg = dict(vocab_size=int(vocab), context_length=int(ctx), width=int(width), layers=nl, heads=heads, embed_dim=int(embed))
        cfg = _lib.TextConfig(int(vocab), int(ctx), int(width), nl, heads, int(embed))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().excel_text_create(C.byref(cfg), C.byref(w), C.byref(self._h)), "excel_text_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().excel_text_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, tokens):
        """tokens [B, context_length] (any integer dtype) -> [B, embed_dim] f32"""
        tok = torch.as_tensor(tokens).to(device=self.device, dtype=torch.int32).contiguous()
        B, ctx = tok.shape
        if ctx != self.cfg["context_length"]:
            raise ValueError(f"tokens must be [B,{self.cfg['context_length']}], got {tuple(tok.shape)}")
        out = torch.empty((B, self.cfg["embed_dim"]), dtype=torch.float32, device=self.device)
        need = lib().excel_text_workspace_bytes(self._h, B)
        ws = _ws(need, self.device)
        check(lib().excel_text_encode(self._h, _p(tok, torch.int32), B, _p(out), _p(ws, torch.uint8), need, _stream()), "excel_text_encode")
        return out


def prompt_ensemble(class_embeddings):
    """clip/clip.py:262-266: [n,E] -> unit-norm mean of the unit-norm rows [E]."""
    e = f32c(class_embeddings)
    n, E = e.shape
    out = torch.empty((E,), dtype=torch.float32, device=e.device)
    check(lib().excel_prompt_ensemble(_p(e), n, E, _p(out), _stream()), "excel_prompt_ensemble")
    return out


def train_losses(seg, attn_pred, pseudo_u8, radius=8, ignore_index=255, w_seg=1.0, w_diver=0.1, aff_labels_u8=None):
    """scripts/train_voc.py:202-215 -> (losses [2] = (seg_loss, diver_loss), d_seg, d_attn_pred).
    aff_labels_u8: the map the affinity labels come from (default: the pseudo labels; :210 switches to the seg arg-max later on)."""
    seg, attn_pred = f32c(seg), f32c(attn_pred)
    B, nc, gh, gw = seg.shape
    H, W = pseudo_u8.shape[-2:]
    losses = torch.empty((2,), dtype=torch.float32, device=seg.device)
    d_seg, d_ap = torch.empty_like(seg), torch.empty_like(attn_pred)
    ws = _ws(lib().excel_train_losses_workspace_bytes(B, nc, H, W), seg.device)
    check(lib().excel_train_losses(_p(seg), _p(attn_pred), _p(pseudo_u8.contiguous(), torch.uint8),
                                   _p(aff_labels_u8.contiguous(), torch.uint8) if aff_labels_u8 is not None else None, B, nc, gh, gw, H, W, radius, ignore_index,
                                   float(w_seg), float(w_diver), _p(losses), _p(d_seg), _p(d_ap), _p(ws, torch.uint8), _stream()), "excel_train_losses")
    return losses, d_seg, d_ap


def lam_to_label(cam, cls_label, img_box=None, bkg_thre=0.5, high_thre=None, low_thre=None, ignore_mid=False, ignore_index=255):
    """utils/camutils.py:123-145 -> (valid_cam [B,F,H,W], pseudo_label uint8 [B,H,W])."""
    cam, cls_label = f32c(cam), f32c(cls_label)
    B, F_, H, W = cam.shape
    valid = torch.empty_like(cam)
    lab = torch.empty((B, H, W), dtype=torch.uint8, device=cam.device)
    box = None if img_box is None else torch.as_tensor(img_box).to(device=cam.device, dtype=torch.int32).contiguous()
    check(lib().excel_lam_to_label(_p(cam), _p(cls_label), _p(box, torch.int32), B, F_, H, W, float(bkg_thre), float(high_thre or 0.0),
                                   float(low_thre or 0.0), 1 if ignore_mid else 0, int(ignore_index), _p(valid), _p(lab, torch.uint8), _stream()),
          "excel_lam_to_label")
    return valid, lab


def normalize_img_u8(hwc_u8, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
    """datasets/transforms.normalize_img + HWC->CHW on the device: uint8 [B,H,W,3] -> f32 [B,3,H,W] (3 B/pixel over PCIe instead of 12)."""
    x = hwc_u8.contiguous()
    B, H, W, Cc = x.shape
    assert Cc == 3 and x.dtype == torch.uint8
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    m, s = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)
    check(lib().excel_normalize_img_u8(_p(x, torch.uint8), B, H, W, m, s, _p(out), _stream()), "excel_normalize_img_u8")
    return out


def denormalize_img(imgs, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), as_float=False):
    """utils/imutils.py:11-25: [B,3,H,W] normalised f32 -> uint8 image (denormalize_img) or that / 255 as f32 (denormalize_img2)."""
    imgs = f32c(imgs)
    B, Cc, H, W = imgs.shape
    assert Cc == 3
    out = torch.empty((B, 3, H, W), dtype=torch.float32 if as_float else torch.uint8, device=imgs.device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(lib().excel_denormalize_img(_p(imgs), B, H, W, m, s, None if as_float else _p(out, torch.uint8), _p(out) if as_float else None,
                                      _stream()), "excel_denormalize_img")
    return out


def seg_scale_accumulate(segs, acc, H, W, flip_mean, init, scale=1.0):
    """tools/infer_seg_voc.py:66-82 for one scale: segs [2B,nc,h,w] -> acc [B,nc,H,W] (allocated when None)."""
    segs = f32c(segs)
    B2, nc, h, w = segs.shape
    B = B2 // 2
    if acc is None:
        acc = torch.empty((B, nc, H, W), dtype=torch.float32, device=segs.device)
        init = True
    check(lib().excel_seg_scale_accumulate(_p(segs), _p(acc), B, nc, h, w, H, W, 1 if flip_mean else 0, 1 if init else 0, float(scale),
                                           _stream()), "excel_seg_scale_accumulate")
    return acc


# ------------------------------------------------------------------ CAM
def clip_feature_surgery(image_features, text_features, num_fg=None, t=2.0, want_full=True):
    """image_features [B,N,C], text_features [T,C] -> (full [B,N,T] | None, slice [B,N-1,F] | None)."""
    image_features = f32c(image_features)
    text_features = f32c(text_features)
    B, N, Cc = image_features.shape
    T = text_features.shape[0]
    dev = image_features.device
    full = torch.empty((B, N, T), dtype=torch.float32, device=dev) if want_full else None
    F_ = T if num_fg is None else num_fg
    sl = torch.empty((B, N - 1, F_), dtype=torch.float32, device=dev) if num_fg is not None else None
    ws = _ws(lib().excel_cam_workspace_bytes(B, N, T), dev)
    check(lib().excel_clip_feature_surgery(_p(image_features), _p(text_features), B, N, Cc, T, F_, float(t), _p(full), _p(sl),
                                           _p(ws, torch.uint8), _stream()), "excel_clip_feature_surgery")
    return full, sl


def patch_text_cam(x_raw, text_features, num_fg=None, t=2.0, want_full=False, want_features=False, mode="bf16x3"):
    """Fused path (excel_patch_text_cam): un-normalised token features x_raw [B,N,C] (VitHandle.forward(want_raw=True)) + text [T,C]
    -> (full [B,N,T] | None, slice [B,N-1,F] | None, image_features [B,N,C] | None): clip.py:353 + :288-310 (column-norm pass, similarity tiles, finish)."""
    x_raw = f32c(x_raw)
    text_features = f32c(text_features)
    B, N, Cc = x_raw.shape
    T = text_features.shape[0]
    dev = x_raw.device
    full = torch.empty((B, N, T), dtype=torch.float32, device=dev) if want_full else None
    F_ = T if num_fg is None else num_fg
    sl = torch.empty((B, N - 1, F_), dtype=torch.float32, device=dev) if num_fg is not None else None
    feats = torch.empty_like(x_raw) if want_features else None
    if full is None and sl is None:
        raise ValueError("patch_text_cam: nothing to compute (want_full=False and num_fg=None)")
    ws = _ws(lib().excel_patch_text_cam_workspace_bytes(B, N, Cc, T), dev)
    check(lib().excel_patch_text_cam(_p(x_raw), _p(text_features), B, N, Cc, T, F_, float(t), GEMM_MODES["f16x3" if mode == "f16x2" else mode], _p(full), _p(sl),
                                     _p(feats), _p(ws, torch.uint8), _stream()), "excel_patch_text_cam")
    return full, sl, feats


# ------------------------------------------------------------------ affinity random walk
def attn_layer_mean(attn, n_layers=6):
    """attn [Lw,B,N,N] -> mean of the last n_layers of attn[:, :, 1:, 1:]  -> [B,P,P]"""
    attn = f32c(attn)
    Lw, B, N, _ = attn.shape
    n = min(n_layers, Lw)
    out = torch.empty((B, N - 1, N - 1), dtype=torch.float32, device=attn.device)
    check(lib().excel_attn_layer_mean(_p(attn), Lw, B, N, Lw - n, n, _p(out), _stream()), "excel_attn_layer_mean")
    return out


def attn_select_mean(attn, seg_attn, n_layers=6):
    """seg_attn branch of refine_cams_with_aff (affutils.py:182-195): attn [Lw,B,N,N], seg_attn [B,P,P] -> [B,P,P]."""
    attn = f32c(attn)
    seg_attn = f32c(seg_attn)
    Lw, B, N, _ = attn.shape
    n = min(n_layers, Lw)
    if tuple(seg_attn.shape) != (B, N - 1, N - 1):
        raise ValueError(f"seg_attn must be [B,P,P] = {(B, N - 1, N - 1)}, got {tuple(seg_attn.shape)}")
    out = torch.empty((B, N - 1, N - 1), dtype=torch.float32, device=attn.device)
    ws = _ws(lib().excel_attn_select_workspace_bytes(B, n), attn.device)
    check(lib().excel_attn_select_mean(_p(attn), Lw, B, N, Lw - n, n, _p(seg_attn), _p(out), _p(ws, torch.uint8), _stream()),
          "excel_attn_select_mean")
    return out


def feature_affinity(feats, mode, beta=1.0, gamma=3.0):
    """feats [B,C,g,g] | [B,C,P] -> [B,P,P].  mode "sigmoid": attn_pred (model_excel.py:70-76);
    mode "mask_softmax": ex_attn of the LVC branch (clip_surgery_model.py:128-137)."""
    feats = f32c(feats)
    feats = feats.reshape(feats.shape[0], feats.shape[1], -1)
    B, Cc, P = feats.shape
    out = torch.empty((B, P, P), dtype=torch.float32, device=feats.device)
    ws = _ws(lib().excel_feature_affinity_workspace_bytes(B, Cc, P), feats.device)
    check(lib().excel_feature_affinity(_p(feats), B, Cc, P, float(beta), float(gamma), {"sigmoid": 0, "mask_softmax": 1}[mode],
                                       _p(out), _p(ws, torch.uint8), _stream()), "excel_feature_affinity")
    return out


def compute_trans_mat(w_aff):
    """[B,P,P] (or [P,P]) -> trans_mat, same shape (utils/affutils.py:8-24)."""
    single = w_aff.dim() == 2
    w = f32c(w_aff[None] if single else w_aff)
    B, P, _ = w.shape
    out = torch.empty_like(w)
    ws = _ws(lib().excel_trans_mat_workspace_bytes(B, P), w.device)
    check(lib().excel_compute_trans_mat(_p(w), B, P, _p(out), _p(ws, torch.uint8), _stream()), "excel_compute_trans_mat")
    return out[0] if single else out


def cls_compact(onehot, smax, want_nchan=False):
    """one-hot [B,F] -> (cls_idx [B,smax] int32, ncls [B] int32[, nchan = ncls+1])."""
    onehot = f32c(onehot)
    B, F_ = onehot.shape
    idx = torch.empty((B, smax), dtype=torch.int32, device=onehot.device)
    n = torch.empty((B,), dtype=torch.int32, device=onehot.device)
    nch = torch.empty((B,), dtype=torch.int32, device=onehot.device) if want_nchan else None
    check(lib().excel_cls_compact(_p(onehot), B, F_, smax, _p(idx, torch.int32), _p(n, torch.int32), _p(nch, torch.int32),
                                  _stream()), "excel_cls_compact")
    return (idx, n, nch) if want_nchan else (idx, n)


def scoremap_box_mask(attr, cls_idx, ncls, g, caa_thre=0.79, want_mask=False):
    attr = f32c(attr)
    B, P, F_ = attr.shape
    smax = cls_idx.shape[1]
    v = torch.zeros((B, smax, P), dtype=torch.float32, device=attr.device)
    m = torch.zeros((B, smax, P), dtype=torch.uint8, device=attr.device) if want_mask else None
    check(lib().excel_scoremap_box_mask(_p(attr), _p(cls_idx, torch.int32), _p(ncls, torch.int32), B, g, F_, smax, float(caa_thre),
                                        _p(v), _p(m, torch.uint8), _stream()), "excel_scoremap_box_mask")
    return v, m


def refine_cams_with_aff_batched(attr, w_aff, cls_idx, ncls, g, caa_thre=0.79):
    """attr [B,P,F], w_aff [B,P,P] -> refined [B,Smax,P] (rows >= ncls[b] are zero)."""
    attr = f32c(attr)
    w_aff = f32c(w_aff)
    B, P, F_ = attr.shape
    smax = cls_idx.shape[1]
    out = torch.zeros((B, smax, P), dtype=torch.float32, device=attr.device)
    ws = _ws(lib().excel_refine_workspace_bytes(B, P, smax), attr.device)
    check(lib().excel_refine_cams_with_aff(_p(attr), _p(w_aff), _p(cls_idx, torch.int32), _p(ncls, torch.int32), B, g, F_, smax,
                                           float(caa_thre), _p(out), _p(ws, torch.uint8), _stream()), "excel_refine_cams_with_aff")
    return out


def _out(out, shape, dtype, device, zero=False):
    """A caller-provided output buffer (the pipeline keeps its buffers across steps: no allocator traffic, no fill launches) or a fresh one."""
    if out is not None:
        if tuple(out.shape) != tuple(shape) or out.dtype != dtype or not out.is_contiguous():
            raise ValueError(f"out= must be a contiguous {dtype} tensor of shape {tuple(shape)}, got {out.dtype} {tuple(out.shape)}")
        return out
    return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)


def cam_upsample_bkg(refined, ncls, g, H, W, out=None, zero_unused=True):
    """refined [B,Smax,P] -> cams [B,Smax+1,H,W] (channel 0 background; channels > ncls[b] are zero unless zero_unused=False:
    nothing on the path reads them)."""
    refined = f32c(refined)
    B, smax, P = refined.shape
    cams = _out(out, (B, smax + 1, H, W), torch.float32, refined.device)
    ws = _ws(B * smax * P * 4, refined.device)
    check(lib().excel_cam_upsample_bkg(_p(refined), _p(ncls, torch.int32), B, g, smax, H, W, _p(cams), _p(ws, torch.uint8),
                                       1 if zero_unused else 0, _stream()), "excel_cam_upsample_bkg")
    return cams


# ------------------------------------------------------------------ PAR / labels / metric
def par_forward(imgs, masks, dilations=PAR_DILATIONS, num_iter=20, nchan=None, w1=0.3, w2=0.01, stream_affinities=False, out=None, ws=None):
    """stream_affinities: stream the 8*ndil affinity planes instead of recomputing them per step (bit-identical outputs; the
    reference form of the recomputing kernel, and what shapes / dilation sets outside its envelope use anyway).
    Channels >= nchan[b] of `out` are not written (zero in a fresh tensor)."""
    imgs = f32c(imgs)
    masks = f32c(masks)
    B, Cmax, H, W = masks.shape
    assert imgs.shape[0] == B and imgs.shape[1] == 3
    h, w = imgs.shape[-2:]
    out = _out(out, masks.shape, torch.float32, masks.device, zero=nchan is not None)
    dil = (C.c_int32 * len(dilations))(*dilations)
    need = lib().excel_par_workspace_bytes(B, Cmax, H, W, len(dilations))
    if ws is None or ws.numel() < need:
        ws = _ws(need, masks.device)
    check(lib().excel_par_forward(_p(imgs), h, w, _p(masks), _p(nchan, torch.int32), B, Cmax, H, W, dil, len(dilations), num_iter,
                                  w1, w2, _p(out), _p(ws, torch.uint8), 1 if stream_affinities else 0, _stream()), "excel_par_forward")
    return out


# ------------------------------------------------------------------ ragged batches (images of different label sizes in one launch)
class RaggedPlan:
    """Tile map + offsets of a batch of images with different (H_b, W_b) (include/excel_hip.h, "ragged batches").
    `hw`: sequence of (H, W).  The table is built on the host by the library (excel_ragged_plan) and copied to the device once."""

    def __init__(self, hw, device):
        import numpy as np
        hw = np.ascontiguousarray(np.asarray(hw, np.int32).reshape(-1, 2))
        self.hw = hw
        self.B = int(hw.shape[0])
        self.info = _lib.RaggedInfo()
        ptr = hw.ctypes.data_as(C.POINTER(C.c_int32))
        check(lib().excel_ragged_plan(ptr, self.B, C.byref(self.info), None), "excel_ragged_plan")
        table = np.empty(int(self.info.table_ints), np.int32)
        check(lib().excel_ragged_plan(ptr, self.B, C.byref(self.info), table.ctypes.data_as(C.POINTER(C.c_int32))), "excel_ragged_plan")
        self.table_host = table
        rec = table[:8 * (self.B + 1)].reshape(self.B + 1, 8)
        self.poff = rec[:, 2].astype(np.int64)          # [B+1] element offsets of the images in a one-plane pitched tensor
        self.loff = rec[:, 4].astype(np.int64)          # [B+1] pixel offsets in a tight u8 map
        self.total_pix = int(self.info.total_pix)
        self.total_label_pix = int(self.info.total_label_pix)
        self.total_tiles = int(self.info.total_tiles)
        self.table = torch.from_numpy(table).to(device, non_blocking=True) if device is not None else None

    def planes(self, packed, b, K):
        """View of image b of a packed K-plane pitched tensor: [K, H_b, W_b] (the row padding sliced off)."""
        H, W = int(self.hw[b, 0]), int(self.hw[b, 1])
        Wp = (W + 3) // 4 * 4
        o = K * int(self.poff[b])
        return packed[o:o + K * H * Wp].view(K, H, Wp)[:, :, :W]

    def label(self, packed_u8, b):
        H, W = int(self.hw[b, 0]), int(self.hw[b, 1])
        o = int(self.loff[b])
        return packed_u8[o:o + H * W].view(H, W)


def normalize_resize_u8_ragged(hwc_packed, plan, S, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), out=None):
    """decoded uint8 HWC images of different sizes, packed back to back -> normalised, bilinearly resized network input [B,3,S,S]
    (datasets/transforms.normalize_img + tools/infer_lam.py:74)."""
    if hwc_packed.dtype != torch.uint8 or hwc_packed.numel() != 3 * plan.total_label_pix:
        raise ValueError(f"hwc_packed must hold {3 * plan.total_label_pix} uint8 values")
    out = _out(out, (plan.B, 3, S, S), torch.float32, hwc_packed.device)
    m, s = (C.c_double * 3)(*mean), (C.c_double * 3)(*std)
    check(lib().excel_normalize_resize_u8_ragged(_p(hwc_packed, torch.uint8), _p(plan.table, torch.int32), plan.B, S, m, s, _p(out), _stream()),
          "excel_normalize_resize_u8_ragged")
    return out


def cam_upsample_bkg_ragged(refined, ncls, g, plan, out=None, zero_unused=True):
    """refined [B,Smax,P] -> packed cams: (Smax+1) pitched planes per image at its own (H_b, W_b)."""
    refined = f32c(refined)
    B, smax, P = refined.shape
    assert B == plan.B
    cams = _out(out, ((smax + 1) * plan.total_pix,), torch.float32, refined.device)
    ws = _ws(B * smax * P * 4, refined.device)
    check(lib().excel_cam_upsample_bkg_ragged(_p(refined), _p(ncls, torch.int32), _p(plan.table, torch.int32), C.byref(plan.info), g, smax,
                                              _p(cams), _p(ws, torch.uint8), 1 if zero_unused else 0, _stream()), "excel_cam_upsample_bkg_ragged")
    return cams


def par_forward_ragged(imgs, masks, plan, Cmax, dilations=PAR_DILATIONS, num_iter=20, nchan=None, w1=0.3, w2=0.01, out=None, ws=None):
    """imgs [B,3,h,w] (uniform), masks = Cmax pitched planes per image -> refined masks, same layout."""
    imgs = f32c(imgs)
    assert imgs.shape[0] == plan.B and imgs.shape[1] == 3 and masks.numel() == Cmax * plan.total_pix
    h, w = imgs.shape[-2:]
    out = _out(out, masks.shape, torch.float32, masks.device, zero=True)
    dil = (C.c_int32 * len(dilations))(*dilations)
    need = lib().excel_par_ragged_workspace_bytes(plan.total_pix, Cmax)
    if ws is None or ws.numel() < need:
        ws = _ws(need, masks.device)
    check(lib().excel_par_forward_ragged(_p(imgs), h, w, _p(masks), _p(nchan, torch.int32), _p(plan.table, torch.int32), C.byref(plan.info), Cmax,
                                         dil, len(dilations), num_iter, w1, w2, _p(out), _p(ws, torch.uint8), _stream()), "excel_par_forward_ragged")
    return out


def argmax_label_ragged(cams, plan, Cmax, nchan=None, cls_idx=None, out=None):
    """packed cams (Cmax pitched planes per image) -> tight uint8 labels [sum H_b*W_b], valid_key lookup applied."""
    smax = cls_idx.shape[1] if cls_idx is not None else Cmax - 1
    lab = _out(out, (plan.total_label_pix,), torch.uint8, cams.device)
    check(lib().excel_argmax_label_ragged(_p(cams), _p(nchan, torch.int32), _p(cls_idx, torch.int32), _p(plan.table, torch.int32), C.byref(plan.info),
                                          smax, Cmax, _p(lab, torch.uint8), _stream()), "excel_argmax_label_ragged")
    return lab


def argmax_label(cams, nchan=None, cls_idx=None, want_i64=False):
    """cams [B,Cmax,H,W] -> labels u8 [B,H,W] (and int64 copy if asked), valid_key lookup applied."""
    cams = f32c(cams)
    B, Cmax, H, W = cams.shape
    smax = cls_idx.shape[1] if cls_idx is not None else Cmax - 1
    l8 = torch.empty((B, H, W), dtype=torch.uint8, device=cams.device)
    l64 = torch.empty((B, H, W), dtype=torch.int64, device=cams.device) if want_i64 else None
    check(lib().excel_argmax_label(_p(cams), _p(nchan, torch.int32), _p(cls_idx, torch.int32), B, smax, Cmax, H * W,
                                   _p(l8, torch.uint8), _p(l64, torch.int64), _stream()), "excel_argmax_label")
    return (l8, l64) if want_i64 else l8


def confusion_accumulate(gt_u8, pred_u8, num_classes, hist=None):
    """hist (int64 [nc,nc], device) += bincount(nc*gt + pred) over gt < nc."""
    gt = gt_u8.contiguous().view(-1)
    pr = pred_u8.contiguous().view(-1)
    assert gt.numel() == pr.numel()
    if hist is None:
        hist = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=gt.device)
    check(lib().excel_confusion_accumulate(_p(gt, torch.uint8), _p(pr, torch.uint8), gt.numel(), num_classes,
                                           _p(hist, torch.int64), _stream()), "excel_confusion_accumulate")
    return hist


# ------------------------------------------------------------------ one-time / auxiliary
def attr_aggregate(text_features, bank, num_fg, topK=0.9):
    """text [T,C], bank [C,K] -> text_attr [C,T] (model/load_attr.py:86-119)."""
    text_features = f32c(text_features)
    bank = f32c(bank)
    T, Cc = text_features.shape
    K = bank.shape[1]
    out = torch.empty((Cc, T), dtype=torch.float32, device=text_features.device)
    check(lib().excel_attr_aggregate(_p(text_features), _p(bank), num_fg, T, Cc, K, float(topK), _p(out), _stream()),
          "excel_attr_aggregate")
    return out


def bilinear_resize(x, H, W, align_corners=False):
    """x [..., h, w] -> [..., H, W]  (F.interpolate mode='bilinear')."""
    x = f32c(x)
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    out = torch.empty(x.shape[:-2] + (H, W), dtype=torch.float32, device=x.device)
    check(lib().excel_bilinear_resize(_p(x), _p(out), planes, h, w, H, W, 1 if align_corners else 0, _stream()),
          "excel_bilinear_resize")
    return out


def pos_embed_resize(pos, g):
    pos = f32c(pos)
    side = int(round((pos.shape[0] - 1) ** 0.5))
    D = pos.shape[1]
    out = torch.empty((g * g + 1, D), dtype=torch.float32, device=pos.device)
    check(lib().excel_pos_embed_resize(_p(pos), side, g, D, _p(out), _stream()), "excel_pos_embed_resize")
    return out


def flip_max_normalize(attr, g):
    """attr [2B,P,F] (second half from flipped inputs) -> [B,P,F] (utils/camutils.py:21-26)."""
    attr = f32c(attr)
    B2, P, F_ = attr.shape
    out = torch.empty((B2 // 2, P, F_), dtype=torch.float32, device=attr.device)
    check(lib().excel_flip_max_normalize(_p(attr), _p(out), B2 // 2, g, F_, _stream()), "excel_flip_max_normalize")
    return out


def lam_scale_accumulate(maps, acc, g, H, W, init):
    """maps [2B,P,F] of one scale -> resize to (H,W), flip-max, (+)= into acc [B,F,H,W]."""
    maps = f32c(maps)
    B2, P, F_ = maps.shape
    if acc is None:
        acc = torch.empty((B2 // 2, F_, H, W), dtype=torch.float32, device=maps.device)
        init = True
    check(lib().excel_lam_scale_accumulate(_p(maps), _p(acc), B2 // 2, g, F_, H, W, 1 if init else 0, _stream()),
          "excel_lam_scale_accumulate")
    return acc


def plane_minmax_normalize_(lam):
    """in place: lam -= min_hw ; lam /= max_hw + 1e-5 for every [..., H, W] plane."""
    H, W = lam.shape[-2:]
    check(lib().excel_plane_minmax_normalize(_p(lam), lam.numel() // (H * W), H * W, _stream()), "excel_plane_minmax_normalize")
    return lam


# ------------------------------------------------------------------ live kernel timing (HIP events on the launch stream)
def prof_enable(on=True, categories=None, every=1):
    """categories: iterable of category names to bracket (None = all); every: bracket every n-th launch of a category
    (each event pair costs ~10 us of GPU idle time, so the timed region samples)."""
    check(lib().excel_prof_set_sampling(int(every)), "excel_prof_set_sampling")
    mask = (1 << 64) - 1
    if categories is not None:
        names = [lib().excel_prof_category_name(i).decode() for i in range(lib().excel_prof_num_categories())]
        mask = 0
        for c in categories:
            mask |= 1 << names.index(c)
    check(lib().excel_prof_set_mask(mask), "excel_prof_set_mask")
    check(lib().excel_prof_enable(1 if on else 0), "excel_prof_enable")


def dcrf_inference(image_u8, prob, iters, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std, is_energy=False):
    """image [H,W,3] uint8, prob [C,H,W] f32 (probabilities, or unary energies when is_energy) -> Q [C,H,W] (excel_dcrf_inference)."""
    prob = f32c(prob)
    if image_u8.dtype != torch.uint8 or not image_u8.is_contiguous():
        image_u8 = image_u8.to(torch.uint8).contiguous()
    Cn, H, W = prob.shape
    assert tuple(image_u8.shape) == (H, W, 3), (tuple(image_u8.shape), (H, W, 3))
    out = torch.empty_like(prob)
    ws = _ws(lib().excel_dcrf_workspace_bytes(H, W, Cn), prob.device)
    check(lib().excel_dcrf_inference(_p(image_u8, torch.uint8), _p(prob), 1 if is_energy else 0, H, W, Cn, int(iters), float(pos_w), float(pos_xy_std),
                                     float(bi_w), float(bi_xy_std), float(bi_rgb_std), _p(out), _p(ws, torch.uint8), _stream()), "excel_dcrf_inference")
    return out


def prof_collect():
    """-> {category: dict(ms=summed elapsed, launches=count, work=algorithmic FLOPs or 0)} and clears the log."""
    n = lib().excel_prof_num_categories()
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    work = (C.c_double * n)()
    check(lib().excel_prof_collect(ms, cnt, work), "excel_prof_collect")
    return {lib().excel_prof_category_name(i).decode(): dict(ms=ms[i], launches=cnt[i], work=work[i]) for i in range(n)}
