"""ExCEL "surgery" CLIP ViT forward restated in numpy fp32 (oracle; test infrastructure only).

Follows clip/clip_surgery_model.py of the reference:
  Attention.forward                :95-159   (ex_feats is None branch :124-126)
  ResidualAttentionBlock.forward   :313-337
  Transformer.forward              :346-371
  VisionTransformer.reload_self_attn :396-416
  VisionTransformer.forward        :419-448
nn.MultiheadAttention (third-party PyTorch, used at :307 for the un-modified
blocks) is restated from its documented semantics: packed in_proj rows q|k|v,
per-head softmax(q k^T / sqrt(d)), out_proj, weights averaged over heads when
need_weights=True.

Weights are a dict keyed like the reference module's state_dict
("transformer.resblocks.3.attn.in_proj_weight", ...), values float32 numpy.
"""
from dataclasses import dataclass

import numpy as np

from .interp import bilinear_resize


@dataclass
class VitConfig:
    width: int = 768
    layers: int = 12
    heads: int = 12
    patch: int = 16
    out_dim: int = 512
    input_resolution: int = 224   # resolution the stored positional grid was made for
    n_surgery: int = 5            # reload_self_attn(layers=6): range(1, 6) -> last 5 blocks (:399)

    @property
    def head_dim(self):
        return self.width // self.heads


def make_vit_weights(cfg: VitConfig, seed: int = 0, attn_gain: float = 4.0, outliers: bool = False, sharp: float = 1.6):
    """Seeded synthetic weights with the reference module's shapes.

    `outliers=True` turns the benign random net into a CLIP-LIKE STRESS NET (applied on top of the same seeded weights, from a second
    random stream, so `outliers=False` stays bit-identical): a handful of "massive activation" channels (their residual-stream values
    are ~50-100x the rest, written by a few MLP output rows and the class / positional embeddings, as trained CLIP ViTs have),
    heavy-tailed (log-normal) LayerNorm gains, and the q / k projections scaled by `sharp` so that attention rows are peaked.  `sharp`
    sets how ill-conditioned the forward is, measured (ViT-B/16 @448, fp32 oracle against its own float64 run, CAM max-abs): 1.4 ->
    1.3e-5, 1.6 -> 3.4e-5 (peaked rows, the default), 1.8 -> 9e-5, 2.0 -> 1.9e-4 (every head near one-hot: fp32 arithmetic itself is
    marginal).  Real CLIP ViT-B/16 weights are not available offline; this is the closest the parity tests get to their regime.

    RandomState (legacy, stream-stable across numpy versions) so the GPU box
    regenerates bit-identical weights.  q/k/v projections get a gain so that
    the softmaxes are far from uniform (a uniform attention would hide
    indexing mistakes).  Test infrastructure, not a reference restatement.
    """
    rs = np.random.RandomState(seed)
    D, L = cfg.width, cfg.layers
    g = cfg.input_resolution // cfg.patch
    f32 = np.float32

    def rn(*shape, std=1.0):
        return (rs.standard_normal(shape) * std).astype(f32)

    w = {}
    w["conv1.weight"] = rn(D, 3, cfg.patch, cfg.patch, std=(3 * cfg.patch * cfg.patch) ** -0.5)
    w["class_embedding"] = rn(D, std=D ** -0.5)
    w["positional_embedding"] = rn(g * g + 1, D, std=D ** -0.5 * 4)
    for nm in ("ln_pre", "ln_post"):
        w[nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(f32)
        w[nm + ".bias"] = rn(D, std=0.1)
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        for nm in ("ln_1", "ln_2"):
            w[p + nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(D)).astype(f32)
            w[p + nm + ".bias"] = rn(D, std=0.1)
        w[p + "attn.in_proj_weight"] = rn(3 * D, D, std=attn_gain ** 0.5 * D ** -0.5)
        w[p + "attn.in_proj_bias"] = rn(3 * D, std=0.1)
        w[p + "attn.out_proj.weight"] = rn(D, D, std=0.5 * D ** -0.5)
        w[p + "attn.out_proj.bias"] = rn(D, std=0.02)
        w[p + "mlp.c_fc.weight"] = rn(4 * D, D, std=D ** -0.5)
        w[p + "mlp.c_fc.bias"] = rn(4 * D, std=0.1)
        w[p + "mlp.c_proj.weight"] = rn(D, 4 * D, std=0.5 * (4 * D) ** -0.5)
        w[p + "mlp.c_proj.bias"] = rn(D, std=0.02)
    w["proj"] = rn(D, cfg.out_dim, std=D ** -0.5)
    if outliers:
        ro = np.random.RandomState(seed + 7919)
        n_out = 3 + int(ro.randint(0, 4))                                  # 3..6 massive channels
        chans = ro.choice(D, n_out, replace=False)
        gains = ro.uniform(50.0, 100.0, n_out).astype(f32)
        # the embeddings put a large constant into those channels for every token, and a few blocks' MLP output rows keep feeding them
        w["class_embedding"][chans] += gains * (ro.choice([-1.0, 1.0], n_out)).astype(f32) * f32(0.5)
        w["positional_embedding"][:, chans] += (gains * ro.choice([-1.0, 1.0], n_out)).astype(f32)[None, :] * f32(0.5)
        for i in ro.choice(L, max(2, L // 3), replace=False):
            p = f"transformer.resblocks.{int(i)}."
            w[p + "mlp.c_proj.weight"][chans, :] *= gains[:, None]
            w[p + "mlp.c_proj.bias"][chans] *= gains
        # heavy-tailed LayerNorm gains (log-normal, sigma 0.5: a few of 768 gains reach 4-5x) - but the massive channels are read with a
        # small gain, as trained nets do (otherwise every LayerNorm output is those channels alone)
        names = ["ln_pre", "ln_post"] + [f"transformer.resblocks.{i}.{nm}" for i in range(L) for nm in ("ln_1", "ln_2")]
        for nm in names:
            gsc = np.exp(0.5 * ro.standard_normal(D)).astype(f32)
            gsc[chans] = f32(0.05)
            w[nm + ".weight"] = (w[nm + ".weight"] * gsc).astype(f32)
        # sharper attention: near-one-hot rows in the later blocks
        for i in range(L):
            p = f"transformer.resblocks.{i}."
            w[p + "attn.in_proj_weight"][:2 * D] *= f32(sharp)
    return w


# Arithmetic type of the restatement: fp32 like the reference.  `precision(np.float64)` re-runs the SAME code in float64 - used only to
# put a number on the fp32 path's own round-off in ill-conditioned regimes (the CLIP-like stress net), never as the parity target.
F32 = np.float32


class precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global F32
        self.saved, F32 = F32, self.dtype
        return self

    def __exit__(self, *exc):
        global F32
        F32 = self.saved


def layer_norm(x, w, b, eps=1e-5):
    # LayerNorm subclass computing in fp32, eps = torch default (:271-277)
    x = x.astype(F32)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * w + b


def quick_gelu(x):
    # QuickGELU :280-282
    return x * (F32(1) / (F32(1) + np.exp(F32(-1.702) * x)))


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True, dtype=F32)


def resize_pos_embed(pos, new_side):
    """:426-435 (and :407-414): bilinear (align_corners=False) resize of the grid part."""
    side = int((pos.shape[0] - 1) ** 0.5)
    if side == new_side:
        return pos
    D = pos.shape[1]
    grid = pos[1:].reshape(side, side, D).transpose(2, 0, 1)          # [D, side, side]
    grid = bilinear_resize(grid, new_side, new_side, align_corners=False)
    grid = grid.reshape(D, new_side * new_side).T
    return np.concatenate([pos[:1], grid], 0).astype(F32)


def _split_heads(t, h):
    N, D = t.shape
    return t.reshape(N, h, D // h).transpose(1, 0, 2)                 # [h, N, d]


def _qkv(y, p, w, h):
    qkv = y @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"]
    D = y.shape[1]
    return (_split_heads(qkv[:, :D], h), _split_heads(qkv[:, D:2 * D], h), _split_heads(qkv[:, 2 * D:], h))


def _merge_heads(t):
    h, N, d = t.shape
    return t.transpose(1, 0, 2).reshape(N, h * d)


def mha_block_attention(y, p, w, cfg):
    """nn.MultiheadAttention(need_weights=True): returns (out [N,D], head-MEAN weights [N,N])."""
    h = cfg.heads
    q, k, v = _qkv(y, p, w, h)
    scale = F32(cfg.head_dim ** -0.5)
    a = softmax((q * scale) @ k.transpose(0, 2, 1))
    o = _merge_heads(a @ v) @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"]
    return o.astype(F32), a.mean(0, dtype=F32)


def feature_similarity(feats, beta=1.0, gamma=3.0):
    """Shared front of the LVC branch (:130-133) and of attn_pred (model/model_excel.py:71-75): channel-normalised
    token similarity, shifted by beta x the mean over the WHOLE batch tensor and scaled by gamma.
    feats [B,C,g,g] (or [B,C,P]) -> [B,P,P] f32."""
    f = np.asarray(feats, F32)
    f = f.reshape(f.shape[0], f.shape[1], -1)
    nrm = np.sqrt((f * f).sum(1, keepdims=True, dtype=F32))
    f = f / np.maximum(nrm, F32(1e-12))                        # F.normalize(dim=1), eps 1e-12
    sim = np.einsum("bcm,bcn->bmn", f, f).astype(F32)          # :131
    return ((sim - sim.mean(dtype=F32) * F32(beta)) * F32(gamma)).astype(F32)   # :132


def ex_attention(ex_feats):
    """Attention.forward :128-137: negatives of the similarity -> -inf, row softmax.  [B,C,g,g] -> ex_attn [B,P,P]."""
    sim = feature_similarity(ex_feats, 1.0, 3.0)
    sim = np.where(sim < 0, -np.inf, sim).astype(F32)          # :133
    return softmax(sim)                                               # :137 (identical for every head)


def surgery_attention(y, p, w, cfg, ex_attn=None):
    """Attention.forward :95-159 -> (x [N,D], x_ori [N,D], head-SUM of attn_ori [N,N]).
    ex_attn [P,P] (LVC branch :127-141) is added to every head's attn[1:,1:]."""
    h = cfg.heads
    q, k, v = _qkv(y, p, w, h)
    scale = F32(cfg.head_dim ** -0.5)
    attn_ori = softmax((q @ k.transpose(0, 2, 1)) * scale)            # :102-103
    a1 = softmax((q @ q.transpose(0, 2, 1)) * scale)                  # :119
    a2 = softmax((k @ k.transpose(0, 2, 1)) * scale)                  # :120
    a3 = softmax((v @ v.transpose(0, 2, 1)) * scale)                  # :121
    attn = (a1 + a2 + a3) / F32(3)                             # :125 / :139
    if ex_attn is not None:
        attn[:, 1:, 1:] = attn[:, 1:, 1:] + np.asarray(ex_attn, F32)[None]    # :140-141
    attn = attn.sum(0, keepdims=True, dtype=F32)               # :146  head-summed, broadcast on V
    x_ori = _merge_heads(attn_ori @ v)                                # :148
    x = _merge_heads(attn @ v)                                        # :149
    Wo, bo = w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"]
    x = x @ Wo.T + bo                                                 # :151
    x_ori = x_ori @ Wo.T + bo                                         # :152
    return x.astype(F32), x_ori.astype(F32), attn_ori.sum(0, dtype=F32)  # :154


def mlp(x, p, w):
    hdn = quick_gelu(x @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])
    return hdn @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]


def patch_embed(img, w, cfg):
    """conv1 (kernel = stride = patch, no bias) :421-423 -> [P, D] for one image [3,S,S]."""
    ps = cfg.patch
    C, S, _ = img.shape
    g = S // ps
    patches = img.reshape(C, g, ps, g, ps).transpose(1, 3, 0, 2, 4).reshape(g * g, C * ps * ps)
    return (patches @ w["conv1.weight"].reshape(cfg.width, -1).T).astype(F32)


def vit_forward_single(img, w, cfg: VitConfig, pos=None, ex_attn=None, aliased_feats=False):
    """VisionTransformer.forward :419-448 for ONE image [3,S,S].

    Returns x [N,out_dim], attn_weights list(L) of [N,N], all_feats list(L) of [N,D]
    (all_feats are the *clean* per-block original-path outputs by default.  aliased_feats=True reproduces
    what the reference's decoder really receives: the per-block list holds VIEWS that later
    in-place updates mutate before clip.py:356 stacks them --
      * the last single-path block's entry aliases `x`, which every surgery block updates with `x += x_res`
        (:319,:329) and the cls swap overwrites (:442): it ends up as the FINAL new-path x;
      * a surgery block's entry aliases its `x_ori`, which the NEXT block updates with `x_ori += x_ori_res`
        (:317) before re-binding the name (:318): it ends up as x_ori + the next block's attention residual;
      * the last block's entry is clean.)
    """
    L = cfg.layers
    first_surgery = L - cfg.n_surgery
    x = patch_embed(img, w, cfg)
    x = np.concatenate([w["class_embedding"][None, :], x], 0)        # :424
    g = int((x.shape[0] - 1) ** 0.5)
    if pos is None:
        pos = resize_pos_embed(w["positional_embedding"], g)         # :426-435
    x = x + pos
    x = layer_norm(x, w["ln_pre.weight"], w["ln_pre.bias"])          # :438
    x = x.astype(F32)
    x_ori = None
    attn_weights, all_feats = [], []
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        if i < first_surgery:                                        # :333-337
            o, a = mha_block_attention(layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"]), p, w, cfg)
            x = x + o
            x = x + mlp(layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"]), p, w)
            x = x.astype(F32)
            feat = x
        else:
            src = x if x_ori is None else x_ori                      # :323 vs :315
            x_res, x_ori_res, a = surgery_attention(
                layer_norm(src, w[p + "ln_1.weight"], w[p + "ln_1.bias"]), p, w, cfg, ex_attn)
            x_ori = src + x_ori_res                                  # :317 / :326
            if aliased_feats and i > first_surgery:
                all_feats[i - 1] = x_ori.astype(F32).copy()   # Q4: the previous entry saw this in-place add
            x_ori = x_ori + mlp(layer_norm(x_ori, w[p + "ln_2.weight"], w[p + "ln_2.bias"]), p, w)
            x_ori = x_ori.astype(F32)
            x = (x + x_res).astype(F32)                       # :319 / :329  (no FFN on the new path)
            feat = x_ori
        attn_weights.append(a)
        all_feats.append(feat.copy())
    x = x.copy()
    if x_ori is not None:
        x[0] = x_ori[0]                                              # :442
        if aliased_feats and first_surgery >= 1:
            all_feats[first_surgery - 1] = x.astype(F32).copy()   # Q4: aliases the new-path x, incl. the cls swap
    x = layer_norm(x, w["ln_post.weight"], w["ln_post.bias"])        # :445
    x = (x @ w["proj"]).astype(F32)                           # :446
    return x, attn_weights, all_feats


def vit_forward(imgs, w, cfg: VitConfig, ex_feats=None, aliased_feats=False):
    """Batched wrapper: imgs [B,3,S,S] -> x [B,N,out], attn [L,B,N,N], feats [L,B,N,D].
    ex_feats [B,C,g,g]: the LVC branch (the batch-global mean of :132 makes this a batch-level quantity)."""
    g = imgs.shape[-1] // cfg.patch
    pos = resize_pos_embed(w["positional_embedding"], g)
    ex = ex_attention(ex_feats) if ex_feats is not None else None
    xs, attns, feats = [], [], []
    for b in range(imgs.shape[0]):
        x, a, f = vit_forward_single(np.asarray(imgs[b], F32), w, cfg, pos, None if ex is None else ex[b], aliased_feats)
        xs.append(x)
        attns.append(np.stack(a, 0))
        feats.append(np.stack(f, 0))
    return np.stack(xs, 0), np.stack(attns, 1), np.stack(feats, 1)


def reload_self_attn(w, cfg: VitConfig, feat_size, mode="train"):
    """VisionTransformer.reload_self_attn :396-416.  The attention surgery itself is
    a pure re-labelling of weights (qkv/proj copied verbatim, :401-404); what changes
    numbers is that a mode containing 'train' permanently resizes the positional grid
    to feat_size (:407-414)."""
    w = dict(w)
    if "train" in mode:
        w["positional_embedding"] = resize_pos_embed(w["positional_embedding"], feat_size)
    return w
