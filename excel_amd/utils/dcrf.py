"""Mirror of utils/dcrf.py (the reference's wrapper of pydensecrf): the mean-field DenseCRF runs in libexcel_hip (excel_dcrf_inference,
crf.hip: permutohedral-lattice message passing on the device).  Inputs may be numpy arrays (like the reference's CPU stage) or device
tensors; outputs follow the input kind.

  DenseCRF(iter_max, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std)(image [H,W,3] uint8, probmap [C,H,W]) -> Q [C,H,W]     :42-68
  crf_inference(img, probs, t=10, scale_factor=1, labels=21)                                                              :7-24
  crf_inference_label(img, labels, t=10, n_labels=21, gt_prob=0.7) -> label map                                            :26-40
"""
import numpy as np
import torch

from .. import ops


def _dev(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to("cuda", dtype).contiguous(), True
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to("cuda", dtype).contiguous(), False


def _run(image, prob, t, pos_w, pos_xy, bi_w, bi_xy, bi_rgb, is_energy=False):
    img, _ = _dev(image, torch.uint8)
    p, was_tensor = _dev(prob, torch.float32)
    q = ops.dcrf_inference(img, p, t, pos_w, pos_xy, bi_w, bi_xy, bi_rgb, is_energy=is_energy)
    return q if was_tensor else q.cpu().numpy()


def unary_from_labels(labels, n_labels, gt_prob, zero_unsure=True):
    """pydensecrf.utils.unary_from_labels (host side: a [n_labels, H*W] table of three constants)."""
    labels = np.asarray(labels).flatten()
    n_energy = -np.log((1.0 - gt_prob) / (n_labels - 1))
    p_energy = -np.log(gt_prob)
    U = np.full((n_labels, len(labels)), n_energy, dtype="float32")
    U[labels - 1 if zero_unsure else labels, np.arange(U.shape[1])] = p_energy
    if zero_unsure:
        U[:, labels == 0] = -np.log(1.0 / n_labels)
    return U


def crf_inference(img, probs, t=10, scale_factor=1, labels=21):
    return _run(img, probs, t, 3, 3 / scale_factor, 10, 80 / scale_factor, 13)


def crf_inference_label(img, labels, t=10, n_labels=21, gt_prob=0.7):
    h, w = np.asarray(img).shape[:2] if not isinstance(img, torch.Tensor) else img.shape[:2]
    U = unary_from_labels(labels.cpu().numpy() if isinstance(labels, torch.Tensor) else labels, n_labels, gt_prob=gt_prob, zero_unsure=False)
    q = _run(img, U.reshape(n_labels, h, w), t, 3, 3, 10, 50, 5, is_energy=True)
    return q.argmax(0) if isinstance(q, np.ndarray) else q.argmax(0)


class DenseCRF(object):
    def __init__(self, iter_max, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std):
        self.iter_max = iter_max
        self.pos_w = pos_w
        self.pos_xy_std = pos_xy_std
        self.bi_w = bi_w
        self.bi_xy_std = bi_xy_std
        self.bi_rgb_std = bi_rgb_std

    def __call__(self, image, probmap):
        return _run(image, probmap, self.iter_max, self.pos_w, self.pos_xy_std, self.bi_w, self.bi_xy_std, self.bi_rgb_std)
