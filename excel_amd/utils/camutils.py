"""Mirror of the live CAM helpers of utils/camutils.py.

  cure_attr_map        :93-97
  cure_attr_map_flip   :8-30   (ex_fts=False; the ex_feats/LVC variant is SURVEY 8(f) 'next')
  multi_scale_lam      the multi-scale fuse formula of :41-61 in its evident intent:
                       per scale: maps -> bilinear resize to (h,w) -> flip-max; sum over scales; min-max normalise.
"""
import torch

from .. import ops


@torch.no_grad()
def cure_attr_map(model, inputs, ex_feats):
    return model(inputs, ex_feats=ex_feats)


@torch.no_grad()
def cure_attr_map_flip(model, inputs, ex_fts=False, flip=True, raw_fts=None):
    b, c, h, w = inputs.shape
    if not flip:
        return model(inputs, ex_feats=raw_fts)
    if ex_fts:
        raise NotImplementedError("ex_fts=True needs the decoder head (SURVEY 8(f) 'next'); pass ex_fts=False")
    inputs_cat = torch.cat([inputs, inputs.flip(-1)], dim=0)          # :15 (memory plumbing)
    attr = model(inputs_cat)[2]                                        # :20
    return ops.flip_max_normalize(attr, h // 16)                       # :21-26
