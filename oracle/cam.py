"""generate_clip_fts / clip_feature_surgery restated in numpy fp32 (oracle; test infrastructure only).

Follows clip/clip.py of the reference:
  generate_clip_fts      :348-358
  clip_feature_surgery   :288-310 (redundant_feats is None branch)
and model/model_excel.py:57-58 (the slice [:, 1:, :num_classes-1]).
"""
import numpy as np

from .vit import vit_forward, softmax, feature_similarity


def generate_clip_fts(imgs, w, cfg, ex_feats=None):
    """-> image_features [B,N,C] (L2-normalised over dim=1 = the TOKEN axis, clip.py:353),
    attn_weights [L,B,N,N], all_feats [L,B,N,D]."""
    x, attn, feats = vit_forward(imgs, w, cfg, ex_feats)
    norm = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    return (x / norm).astype(np.float32), attn, feats


def clip_feature_surgery(image_features, text_features, t=2):
    """image_features [B,N,C], text_features [T,C] -> attr_maps [B,N,T] (clip.py:295-308).

    Written in the reference's broadcast-multiply-reduce form, image by image.
    """
    out = []
    T = text_features.shape[0]
    for f in image_features:                                           # [N,C]
        prob = f[:1] @ text_features.T                                 # :295
        prob = softmax(prob * np.float32(t), -1)                       # :296
        w = prob / prob.mean(-1, keepdims=True, dtype=np.float32)      # :297
        feats = f[:, None, :] * text_features[None, :, :]              # :301  [N,T,C]
        feats = feats * w.reshape(1, T, 1)                             # :302
        red = feats.mean(1, keepdims=True, dtype=np.float32)           # :303
        feats = feats - red                                            # :304
        sim = feats.sum(-1, dtype=np.float32)                          # :306  [N,T]
        mn = sim.min(0, keepdims=True)
        mx = sim.max(0, keepdims=True)
        out.append((sim - mn) / (mx - mn))                             # :308 (min/max over ALL N tokens)
    return np.stack(out, 0).astype(np.float32)


def attn_pred(attn_fts):
    """model/model_excel.py:70-76: sigmoid of the shifted/scaled channel-normalised similarity of the decoder
    features.  attn_fts [B,C,g,g] -> [B,P,P]."""
    z = feature_similarity(attn_fts, 1.0, 3.0)
    return (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(np.float32)


def attr_maps_raw(imgs, w, cfg, text_attr, num_fg, ex_feats=None):
    """ExCEL_model.forward hot lines (model_excel.py:57-58; :50-53 with ex_feats): text_attr is [C,T]."""
    f, attn, feats = generate_clip_fts(imgs, w, cfg, ex_feats)
    maps = clip_feature_surgery(f, text_attr.T)[:, 1:, :num_fg]
    return maps, attn, f, feats
