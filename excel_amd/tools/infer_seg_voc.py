"""Multi-scale / flip segmentation inference with the decoder head: mirror of tools/infer_seg_voc.py:47-101 (`_validate`).

  per image batch:  resize to (S,S) -> cat(x, flip(x)) -> model(.)[0] = seg logits -> bilinear to the input size
                    scale 1.0: the un-flipped half alone (:69); other scales: (seg + flip(seg_flipped)) / 2 (:79)
                    mean over scales (:82) -> bilinear to the label size (:84) -> arg-max (:85) -> confusion matrix
All tensor work runs in libexcel_hip.so (ViT, decoder head, excel_seg_scale_accumulate, excel_bilinear_resize,
excel_argmax_label, excel_confusion_accumulate); torch is used for device memory and the flip/cat copies.
"""
import torch

from .. import ops


@torch.no_grad()
def multi_scale_seg(model, inputs, resize_size, scales=(1.0, 0.5, 0.75, 1.5), flip_first=False):
    """inputs [B,3,h,w] -> msc_seg [B,nc,h,w] (the tensor the reference stores as {"msc_seg": ...}, :89).
    flip_first=True: the variant of tools/test_msc_flip_voc.py:89-107, where scale 1.0 is flip-averaged as well."""
    if model._dec is None:
        raise RuntimeError("multi_scale_seg needs the decoder head: build ExCEL_model with decoder_state_dict=")
    B, _, h, w = inputs.shape
    todo = [1.0] + [s for s in scales if s != 1.0]                    # :63-70 first, then :72-80
    acc = None
    for i, sc in enumerate(todo):
        S = resize_size if sc == 1.0 else int(resize_size * sc)       # :64 / :74
        x = ops.bilinear_resize(inputs, S, S, align_corners=False)    # :65 / :75
        segs = model(torch.cat([x, x.flip(-1)], dim=0))[0]            # :66-67 / :76-77
        acc = ops.seg_scale_accumulate(segs, acc, h, w, flip_mean=(sc != 1.0 or flip_first), init=(i == 0),
                                       scale=(1.0 / len(todo)) if i == len(todo) - 1 else 1.0)     # :68-70 / :78-80, :82
    return acc


@torch.no_grad()
def seg_labels(msc_seg, label_hw):
    """:84-85 -> uint8 labels [B,H,W] on the device."""
    H, W = int(label_hw[0]), int(label_hw[1])
    resized = ops.bilinear_resize(msc_seg, H, W, align_corners=False)
    return ops.argmax_label(resized)


@torch.no_grad()
def validate_seg(model, batches, num_classes, resize_size, scales=(1.0, 0.5, 0.75, 1.5)):
    """batches: iterable of (inputs [B,3,h,w] device f32, labels [B,H,W] uint8 device).  -> scores dict (evaluate.scores layout)."""
    from ..utils import evaluate
    hist = None
    for inputs, labels in batches:
        pred = seg_labels(multi_scale_seg(model, inputs, resize_size, scales), labels.shape[-2:])
        hist = ops.confusion_accumulate(labels, pred, num_classes, hist)
    return evaluate.scores_from_hist(hist), hist
