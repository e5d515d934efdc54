"""Mirror of the hot functions of utils/affutils.py over the HIP kernels.

  compute_trans_mat            :8-24
  refine_cams_with_aff         :177-223 (both branches; seg_attn needs per-layer maps: model(img, n_attn_out>=6))
  refine_cams_with_bkg_weclip  :161-174  (+ generate_cam_label/scale_cam_image :55-78, _refine_cams :80-89)

The reference hops to the host per class (numpy + OpenCV, :207-217, :59-66); here everything stays on the device:
scoremap2bbox (:26-53) runs as one workgroup per (image, class) inside excel_refine_cams_with_aff.
"""
import numpy as np
import torch

from .. import ops
from ..clip.clip import LazyAttnWeights


def compute_trans_mat(attn_weight):
    return ops.compute_trans_mat(attn_weight)


def _w_aff_of(attn_weights, attn_layers):
    if isinstance(attn_weights, LazyAttnWeights):
        if attn_weights.aff_layers != attn_layers:
            raise ValueError("attention mean was reduced over a different number of layers")
        return attn_weights.w_aff
    aw = attn_weights
    if aw.dim() == 3:                  # [L,N,N] of one image, as tools/infer_lam.py:90 passes it
        aw = aw[:, None]
    return ops.attn_layer_mean(aw, attn_layers)          # mean(attn[-layers:, 1:, 1:])  (:180,:197)


@torch.no_grad()
def refine_cams_with_aff(attr_map, attn_weights, cls_label, size, caa_thre=0.79, attn_layers=6, seg_attn=None):
    """attr_map [P,F], attn_weights [L,N,N] | LazyAttnWeights, cls_label [F] -> (list of k x [g,g], cls_lst cpu int64)."""
    h, w = size
    g = h // 16
    if seg_attn is not None:                                                                  # :182-195
        stacked = attn_weights.stacked if isinstance(attn_weights, LazyAttnWeights) else attn_weights
        if stacked is None or stacked.shape[0] < attn_layers:
            raise ValueError("the seg_attn branch needs the per-layer attention maps: call model(img, n_attn_out=%d)" % attn_layers)
        if stacked.dim() == 3:
            stacked = stacked[:, None]
        w_aff = ops.attn_select_mean(stacked[:, :1].contiguous(), seg_attn.reshape(1, g * (w // 16), -1), attn_layers)
    else:
        w_aff = _w_aff_of(attn_weights, attn_layers)
    cls_label = torch.as_tensor(cls_label)
    cls_lst = torch.where(cls_label.detach().cpu() != 0)[0]                                   # :203
    k = int(cls_lst.numel())
    if k == 0:
        return [], cls_lst
    idx, n = ops.cls_compact(cls_label.reshape(1, -1).float().to(attr_map.device), k)
    refined = ops.refine_cams_with_aff_batched(attr_map.reshape(1, g * g, -1), w_aff[:1], idx, n, g, caa_thre)
    return [refined[0, s].reshape(g, w // 16) for s in range(k)], cls_lst


@torch.no_grad()
def refine_cams_with_bkg_weclip(cam_refined_list, inputs_denorm, cls_lst, par, size):
    """-> (cam_labels [1,H,W] int64, cams [k+1,H,W])   (:161-174)"""
    H, W = int(size[0]), int(size[1])
    k = len(cam_refined_list)
    g = cam_refined_list[0].shape[0]
    dev = cam_refined_list[0].device
    refined = torch.stack([c.reshape(-1) for c in cam_refined_list], 0)[None].contiguous()       # [1,k,P]
    ncls = torch.tensor([k], dtype=torch.int32, device=dev)
    cams = ops.cam_upsample_bkg(refined, ncls, g, H, W)                                           # [1,k+1,H,W]
    out = par(inputs_denorm[None].float(), cams)                                                  # _refine_cams :80-84
    idx = torch.as_tensor(np.asarray(cls_lst), dtype=torch.int32).reshape(1, k).to(dev)
    _, lab = ops.argmax_label(out, None, idx, want_i64=True)                                      # :86-87, :168
    return lab, cams[0]
