"""Mirror of engine/validatation_engine.py:11-51 (the in-training validation pass): per batch model(inputs) -> seg + LAMs,
seg_attn-gated affinity refinement (caa_thre 0.75), PAR labels; two score dicts (pseudo labels, seg predictions).

The tensor work runs in libexcel_hip.so through the mirrored modules; needs a model built with decoder weights
(`ExCEL_model(..., decoder_state_dict=)`), like the reference at that point of training."""
import torch

from .. import ops
from ..tools.infer_lam import format_scores_table
from ..utils import evaluate
from ..utils.affutils import refine_cams_with_aff, refine_cams_with_bkg_weclip


@torch.no_grad()
def build_validation(model=None, par=None, val_loader=None, device="cuda", num_classes=21, resize_size=320, class_list=None):
    """val_loader yields (name, inputs [B,3,h,w], labels [B,H,W], cls_labels [B,F]) like datasets/voc.py.
    -> (table string, attr_aff_score, seg_score)"""
    hist_aff = hist_seg = None
    for _, data in enumerate(val_loader):
        name, inputs, labels, cls_labels = data
        inputs = torch.as_tensor(inputs).to(device).float()
        inputs = ops.bilinear_resize(inputs, resize_size, resize_size, align_corners=False)            # :20
        cls_labels = torch.as_tensor(cls_labels).to(device).float()
        labels_u8 = torch.as_tensor(labels).to(device=device, dtype=torch.uint8)
        segs, fts_diver, attr_maps_raw, attn_weights, attn_pred = model(inputs, n_attn_out=6)           # :25
        if segs is None or attn_pred is None:
            raise RuntimeError("build_validation needs the decoder head (ExCEL_model(..., decoder_state_dict=))")
        resized = ops.bilinear_resize(segs, labels_u8.shape[-2], labels_u8.shape[-1], align_corners=False)   # :27
        for i, attr_map in enumerate(attr_maps_raw):                                                 # :29
            refined, cls_lst = refine_cams_with_aff(attr_map, attn_weights[:, i, ...], cls_labels[i], size=inputs.shape[2:],
                                                    seg_attn=attn_pred[i][None], caa_thre=0.75)       # :33
            lab, _ = refine_cams_with_bkg_weclip(refined, inputs[i], cls_lst, par, labels_u8.shape[-2:])   # :34
            # the reference appends only the LAST image's pseudo label of each batch (:36 sits outside the loop); with its
            # batch size of 1 that is every image, which is what is accumulated here
            hist_aff = evaluate.hist_from_labels([labels_u8[i]], [lab[0]], num_classes, device, hist_aff)
        hist_seg = ops.confusion_accumulate(labels_u8, ops.argmax_label(resized), num_classes, hist_seg)  # :37
    attr_aff_score = evaluate.scores_from_hist(hist_aff)                                            # :40
    seg_score = evaluate.scores_from_hist(hist_seg)                                                 # :41
    cats = class_list or [str(i) for i in range(num_classes)]
    table = "Attr_aff_Pseudo\n" + format_scores_table(attr_aff_score, cats) + "\nSeg_Preds\n" + format_scores_table(seg_score, cats)
    return table, attr_aff_score, seg_score
