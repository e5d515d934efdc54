"""Mirror of the live CAM helpers of utils/camutils.py.

  cure_attr_map        :93-97
  cure_attr_map_flip   :8-30   (ex_fts=True needs model.feature_head, the caller's decoder, to produce ex_feats)
  multi_scale_lam      the multi-scale fuse formula of :41-61 in its evident intent:
                       per scale: maps -> bilinear resize to (h,w) -> flip-max; sum over scales; min-max normalise.
"""
import torch

from .. import ops


@torch.no_grad()
def cure_attr_map(model, inputs, ex_feats):
    return model(inputs, ex_feats=ex_feats)


@torch.no_grad()
def cure_attr_map_flip(model, inputs, ex_fts=True, flip=True, raw_fts=None):
    b, c, h, w = inputs.shape
    if not flip:
        return model(inputs, ex_feats=raw_fts)
    inputs_cat = torch.cat([inputs, inputs.flip(-1)], dim=0)          # :15 (memory plumbing)
    if ex_fts:
        ex_feats = model(inputs_cat)[1]                                # :17
        if ex_feats is None:
            raise RuntimeError("ex_fts=True needs decoder features: set model.feature_head (the learned decoder is outside "
                               "this library) or call with ex_fts=False")
        attr = model(inputs_cat, ex_feats=ex_feats)                    # :18
    else:
        attr = model(inputs_cat)[2]                                    # :20
    return ops.flip_max_normalize(attr, h // 16)                       # :21-26


@torch.no_grad()
def multi_scale_lam(model, inputs, scales=(1.0, 0.5, 0.75, 1.5)):
    """Multi-scale + flip fused LAMs [B,F,h,w] (utils/camutils.py:32-63 `multi_scale_lam2`, written as the evident intent:
    the reference interpolates the 3-D [2b,P,F] tensor directly, which cannot run - SURVEY 8 a16).  Scales follow
    scripts/train_voc.py:53.  Scaled inputs are rounded to a multiple of the patch size (the ViT needs whole patches)."""
    b, c, h, w = inputs.shape
    acc = None
    for s in scales:
        hs, ws = int(s * h) // 16 * 16, int(s * w) // 16 * 16
        x = inputs if (hs, ws) == (h, w) else ops.bilinear_resize(inputs, hs, ws, align_corners=False)     # :50
        x2 = torch.cat([x, x.flip(-1)], dim=0)                                                              # :51
        maps = model(x2)[2]                                                                                  # :53-54
        acc = ops.lam_scale_accumulate(maps, acc, hs // 16, h, w, init=acc is None)                         # :56-59 resize, flip-max, sum
    return ops.plane_minmax_normalize_(acc)                                                                  # :61-63


@torch.no_grad()
def lam_to_label(cam, cls_label, img_box=None, bkg_thre=0.5, high_thre=None, low_thre=None, ignore_mid=False, ignore_index=None):
    """utils/camutils.py:123-145 (labels come back as uint8; 255 = ignore)."""
    return ops.lam_to_label(cam, cls_label, img_box, bkg_thre, high_thre, low_thre, ignore_mid, 255 if ignore_index is None else ignore_index)
