"""The training-free evaluation harness restated on top of the oracle ops
(oracle; test infrastructure only).

Follows tools/infer_lam.py:63-128 (build_validation) of the reference, with the
model call replaced by the three hot lines of ExCEL_model.forward
(model/model_excel.py:56-58).  Batch size is 1 per step exactly like
tools/infer_lam.py:167.
"""
import numpy as np

from . import aff as _aff
from . import cam as _cam
from . import evaluate as _ev
from .interp import bilinear_resize
from .par import PAR


def run_sample(img, cls_label, label_hw, w, cfg, text_attr, num_fg, par, resize_size,
               caa_thre=0.79, return_all=False):
    """img [3,h,w] f32 (already normalised), cls_label [F], label_hw=(H,W).
    Returns label [H,W] int64 (and intermediates when return_all)."""
    inputs = bilinear_resize(np.asarray(img, np.float32)[None], resize_size, resize_size,
                             align_corners=False)                                     # infer_lam.py:74
    maps, attn, feats, _ = _cam.attr_maps_raw(inputs, w, cfg, text_attr, num_fg)       # :79
    refined, cls_lst = _aff.refine_cams_with_aff(maps[0], attn[:, 0], cls_label,
                                                 size=inputs.shape[2:], caa_thre=caa_thre)   # :93
    label, cams = _aff.refine_cams_with_bkg_weclip(refined, inputs[0], cls_lst, par, label_hw)  # :94
    if return_all:
        return dict(inputs=inputs, attr_maps_raw=maps, attn_weights=attn, image_features=feats,
                    refined=refined, cls_lst=cls_lst, cams=cams, label=label[0])
    return label[0]


def build_validation(samples, w, cfg, text_attr, num_classes=21, resize_size=448,
                     dilations=(1, 2, 4, 8, 12, 24), num_iter=20, caa_thre=0.79):
    """samples: iterable of (img [3,h,w], gt [H,W] uint8, cls_label [F]).
    Returns (hist [nc,nc] int64, list of predicted labels)."""
    par = PAR(dilations, num_iter)                                                     # infer_lam.py:168
    gts, preds = [], []
    for img, gt, cls_label in samples:
        lab = run_sample(img, cls_label, gt.shape[-2:], w, cfg, text_attr, num_classes - 1, par,
                         resize_size, caa_thre)
        preds.append(lab.astype(np.int16))                                             # :113
        gts.append(np.asarray(gt).astype(np.int16))                                    # :114
    hist = _ev.hist_of(gts, preds, num_classes)                                        # :121
    return hist, preds
