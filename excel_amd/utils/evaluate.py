"""scores / _fast_hist - mirror of utils/evaluate.py:9-50 with the histogram on the device."""
import numpy as np
import torch

from .. import ops


def hist_from_labels(label_trues, label_preds, num_classes=21, device="cuda", hist=None):
    """Accumulate the [nc,nc] int64 confusion matrix on the GPU (evaluate.py:9-20)."""
    for lt, lp in zip(label_trues, label_preds):
        lt = torch.as_tensor(np.asarray(lt) if not torch.is_tensor(lt) else lt)
        lp = torch.as_tensor(np.asarray(lp) if not torch.is_tensor(lp) else lp)
        # labels are small non-negative ints, 255 = ignore; uint8 is lossless for them
        gt = lt.to(device=device).to(torch.uint8)
        pr = lp.to(device=device).to(torch.uint8)
        hist = ops.confusion_accumulate(gt, pr, num_classes, hist)
    return hist


def scores_from_hist(hist):
    """The metric arithmetic of evaluate.py:21-50 on a 21x21 (81x81) matrix - host-side float64 like the reference."""
    hist = np.asarray(hist.detach().cpu().numpy() if torch.is_tensor(hist) else hist, np.float64)
    n = hist.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
        TP = np.diag(hist)
        FN = hist.sum(axis=1) - TP
        FP = hist.sum(axis=0) - TP
        cr = FP / TP
        precision = TP / (TP + FP)
        recall = TP / (TP + FN)
    return {"pAcc": acc, "mAcc": acc_cls, "miou": mean_iu, "iou": dict(zip(range(n), iu)),
            "confusion": dict(zip(range(n), cr)), "precision": dict(zip(range(n), precision)),
            "recall": dict(zip(range(n), recall))}


def scores(label_trues, label_preds, num_classes=21):
    return scores_from_hist(hist_from_labels(label_trues, label_preds, num_classes))
