"""attr_aggregate (TSE text-bank fusion) restated in numpy fp32 (oracle; test infrastructure only).

Follows model/load_attr.py:86-119 of the reference.
"""
import numpy as np

from .vit import softmax


def attr_aggregate(text_features, attr_features, num_classes, topK=0.9):
    """text_features [T,C] (fg rows first, then bg prompts), attr_features [C,K] bank
    -> text_attr [C,T] (columns unit-norm)."""
    text_features = np.asarray(text_features, np.float32)
    attr_features = np.asarray(attr_features, np.float32)
    fg = text_features[:num_classes]                                   # :95
    bg = text_features[num_classes:]                                   # :96
    logit = fg @ attr_features                                         # :101
    K = attr_features.shape[1]
    topk = int((1 - topK) * K)                                         # :100
    corr = logit.copy()
    if topk > 0:
        # :102-110: sort descending, set the last `topk` to -inf, scatter back
        idx = np.argsort(-logit, axis=-1, kind="stable")
        drop = idx[:, -topk:]
        np.put_along_axis(corr, drop, -np.inf, axis=-1)
    corr = softmax(corr, -1)                                           # :112
    agg = corr @ attr_features.T + fg                                  # :113
    agg = np.concatenate([agg, bg], 0)                                 # :117
    agg = agg / np.sqrt((agg * agg).sum(1, keepdims=True, dtype=np.float32))   # :118
    return agg.T.astype(np.float32).copy()
