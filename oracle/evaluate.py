"""Confusion-matrix mIoU restated in numpy (oracle; test infrastructure only).

Follows utils/evaluate.py of the reference: _fast_hist :9-15, scores :17-50.
"""
import numpy as np


def fast_hist(label_true, label_pred, num_classes):
    label_true = np.asarray(label_true)
    label_pred = np.asarray(label_pred)
    mask = (label_true >= 0) & (label_true < num_classes)                              # :10
    hist = np.bincount(num_classes * label_true[mask].astype(int) + label_pred[mask].astype(int),
                       minlength=num_classes ** 2)                                     # :11-14
    return hist.reshape(num_classes, num_classes)


def hist_of(label_trues, label_preds, num_classes=21):
    hist = np.zeros((num_classes, num_classes), np.int64)
    for lt, lp in zip(label_trues, label_preds):
        hist += fast_hist(np.asarray(lt).flatten(), np.asarray(lp).flatten(), num_classes)
    return hist


def scores_from_hist(hist):
    hist = np.asarray(hist, np.float64)
    n = hist.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()                                         # :21
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))                         # :22-23
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))     # :24
        valid = hist.sum(axis=1) > 0                                                   # :25
        mean_iu = np.nanmean(iu[valid])                                                # :26
        TP = np.diag(hist)                                                             # :30
        FN = hist.sum(axis=1) - TP
        FP = hist.sum(axis=0) - TP
        cr = FP / TP                                                                   # :34
        precision = TP / (TP + FP)                                                     # :37
        recall = TP / (TP + FN)                                                        # :39
    return {
        "pAcc": acc, "mAcc": acc_cls, "miou": mean_iu,
        "iou": dict(zip(range(n), iu)),
        "confusion": dict(zip(range(n), cr)),
        "precision": dict(zip(range(n), precision)),
        "recall": dict(zip(range(n), recall)),
    }


def scores(label_trues, label_preds, num_classes=21):
    return scores_from_hist(hist_of(label_trues, label_preds, num_classes))
