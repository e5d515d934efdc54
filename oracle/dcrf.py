"""Fully-connected CRF mean-field inference (utils/dcrf.py of the reference) restated in numpy (oracle; test infrastructure only).

PARITY UNPINNED: the reference calls pydensecrf (conda pin pydensecrf=1.0rc3, requirements.txt:4), a wrapper of Kraehenbuehl & Koltun's
densecrf ("Efficient Inference in Fully Connected CRFs with Gaussian Edge Potentials", NIPS 2011; permutohedral lattice filtering of Adams,
Baek & Davis 2010).  The library is a third-party dependency absent from /root/reference and from this image, and the reference has no
test or golden vector for this stage, so what is restated here is the PUBLISHED algorithm as that library implements it:

  unary_from_softmax      U = -log(clip(p, 1e-5, 1))                                        (pydensecrf/utils.py)
  DenseCRF2D features     Gaussian: (x/sxy, y/sxy); bilateral: (x/sxy, y/sxy, r/srgb, g/srgb, b/srgb)   (densecrf.cpp addPairwise*)
  Permutohedral.init      elevate with scale_factor_i = sqrt(2/3)(d+1)/sqrt((i+1)(i+2)), round to the nearest 0-coloured lattice point,
                          rank, barycentric weights, the d+1 simplex vertices; blur neighbours key -+ 1 (axis j: +- d)    (permutohedral.cpp)
  Permutohedral.compute   splat (sum of w * value), d+1 blur passes new = old + (n1 + n2)/2, slice with alpha = 1/(1 + 2^-d)
  DenseKernel             NORMALIZE_SYMMETRIC: norm = 1/sqrt(K 1 + 1e-20); message = norm * K(norm * Q)         (pairwise.cpp)
  PottsCompatibility      pairwise term = -w * message                                                          (labelcompatibility.cpp)
  DenseCRF.inference      Q = softmax(-U); repeat: Q = softmax(-U + sum_k w_k message_k(Q))                       (densecrf.cpp)
and utils/dcrf.py:42-68 (class DenseCRF), :7-24 (crf_inference), :26-40 (crf_inference_label).
"""
import numpy as np


class Permutohedral:
    def __init__(self, feature):
        """feature [N,d] float32."""
        f = np.asarray(feature, np.float32)
        N, d = f.shape
        self.N, self.d = N, d
        inv_std = np.float32(np.sqrt(2.0 / 3.0) * (d + 1))
        scale = np.array([np.float32(1.0 / np.sqrt(float((i + 2) * (i + 1)))) * inv_std for i in range(d)], np.float32)
        cf = f * scale[None, :]
        elevated = np.zeros((N, d + 1), np.float32)
        sm = np.zeros(N, np.float32)
        for j in range(d, 0, -1):
            elevated[:, j] = sm - np.float32(j) * cf[:, j - 1]
            sm = sm + cf[:, j - 1]
        elevated[:, 0] = sm
        down = np.float32(1.0 / (d + 1))
        up = np.float32(d + 1)
        v = elevated * down
        upv, dnv = np.ceil(v) * up, np.floor(v) * up
        rem0 = np.where(upv - elevated < elevated - dnv, upv, dnv).astype(np.int32)      # nearest multiple of d+1
        ssum = (rem0 // (d + 1)).sum(1)                                                  # (exact: rem0 is a multiple of d+1)
        diff = elevated - rem0.astype(np.float32)
        rank = np.zeros((N, d + 1), np.int32)
        for i in range(d):
            for j in range(i + 1, d + 1):
                lt = diff[:, i] < diff[:, j]
                rank[:, i] += lt
                rank[:, j] += ~lt
        rank += ssum[:, None]
        lo, hi = rank < 0, rank > d
        rank = np.where(lo, rank + d + 1, np.where(hi, rank - (d + 1), rank))
        rem0 = np.where(lo, rem0 + d + 1, np.where(hi, rem0 - (d + 1), rem0))
        bary = np.zeros((N, d + 2), np.float32)
        rows = np.arange(N)
        for i in range(d + 1):
            vv = (elevated[:, i] - rem0[:, i].astype(np.float32)) * down
            np.add.at(bary, (rows, d - rank[:, i]), vv)
            np.add.at(bary, (rows, d - rank[:, i] + 1), -vv)
        bary[:, 0] += np.float32(1.0) + bary[:, d + 1]
        canonical = np.zeros((d + 1, d + 1), np.int32)
        for i in range(d + 1):
            canonical[i, :d - i + 1] = i
            canonical[i, d - i + 1:] = i - (d + 1)
        keys = np.zeros((N, d + 1, d), np.int32)
        for r in range(d + 1):
            keys[:, r, :] = rem0[:, :d] + canonical[r][rank[:, :d]]
        flat = keys.reshape(N * (d + 1), d)
        uniq, inv = np.unique(flat, axis=0, return_inverse=True)
        self.M = uniq.shape[0]
        self.offset = inv.reshape(N, d + 1).astype(np.int64)
        self.bary = bary[:, :d + 1].copy()
        look = {tuple(k): i for i, k in enumerate(uniq.tolist())}
        self.n1 = np.full((d + 1, self.M), -1, np.int64)
        self.n2 = np.full((d + 1, self.M), -1, np.int64)
        for j in range(d + 1):
            a, b = uniq - 1, uniq + 1
            if j < d:
                a[:, j] = uniq[:, j] + d
                b[:, j] = uniq[:, j] - d
            self.n1[j] = [look.get(tuple(k), -1) for k in a.tolist()]
            self.n2[j] = [look.get(tuple(k), -1) for k in b.tolist()]

    def compute(self, values):
        """values [N,C] -> filtered [N,C] (float32)."""
        v = np.asarray(values, np.float32)
        C = v.shape[1]
        lat = np.zeros((self.M + 1, C), np.float32)                      # row 0 = the "absent neighbour" zero row
        for j in range(self.d + 1):
            np.add.at(lat, self.offset[:, j] + 1, self.bary[:, j:j + 1] * v)
        for j in range(self.d + 1):
            new = lat.copy()
            new[1:] = lat[1:] + np.float32(0.5) * (lat[self.n1[j] + 1] + lat[self.n2[j] + 1])
            new[0] = 0
            lat = new
        alpha = np.float32(1.0 / (1.0 + 2.0 ** (-self.d)))
        out = np.zeros_like(v)
        for j in range(self.d + 1):
            out += self.bary[:, j:j + 1] * lat[self.offset[:, j] + 1] * alpha
        return out


class DenseKernel:
    """DIAG_KERNEL, NORMALIZE_SYMMETRIC (the pydensecrf defaults the reference relies on)."""

    def __init__(self, feature):
        self.lat = Permutohedral(feature)
        ones = self.lat.compute(np.ones((feature.shape[0], 1), np.float32))[:, 0]
        self.norm = (1.0 / np.sqrt(ones + np.float32(1e-20))).astype(np.float32)

    def apply(self, Q):
        return self.lat.compute(Q * self.norm[:, None]) * self.norm[:, None]


def _softmax_rows(x):
    m = x.max(1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def unary_from_softmax(sm, clip=1e-5):
    C = sm.shape[0]
    return (-np.log(np.clip(np.asarray(sm, np.float32), clip, 1.0))).reshape(C, -1).astype(np.float32)


def unary_from_labels(labels, n_labels, gt_prob, zero_unsure=True):
    labels = np.asarray(labels).flatten()
    n_energy = -np.log((1.0 - gt_prob) / (n_labels - 1))
    p_energy = -np.log(gt_prob)
    U = np.full((n_labels, len(labels)), n_energy, dtype="float32")
    U[labels - 1 if zero_unsure else labels, np.arange(U.shape[1])] = p_energy
    if zero_unsure:
        U[:, labels == 0] = -np.log(1.0 / n_labels)
    return U


def features_2d(H, W, sxy, rgb=None, srgb=None):
    ys, xs = np.mgrid[0:H, 0:W]
    f = [xs.reshape(-1).astype(np.float32) / np.float32(sxy), ys.reshape(-1).astype(np.float32) / np.float32(sxy)]   # float(i) / sx (densecrf.cpp)
    if rgb is not None:
        im = np.asarray(rgb).reshape(-1, 3).astype(np.float32)
        f += [im[:, 0] / np.float32(srgb), im[:, 1] / np.float32(srgb), im[:, 2] / np.float32(srgb)]
    return np.stack(f, 1).astype(np.float32)


def dense_crf_2d(image, unary, iters, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std):
    """image [H,W,3] uint8, unary [C, H*W] -> Q [C,H,W]   (DenseCRF2D + addPairwiseGaussian + addPairwiseBilateral + inference)."""
    H, W = image.shape[:2]
    C = unary.shape[0]
    U = np.asarray(unary, np.float32).T                                  # [N,C]
    kg = DenseKernel(features_2d(H, W, pos_xy_std))
    kb = DenseKernel(features_2d(H, W, bi_xy_std, image, bi_rgb_std))
    Q = _softmax_rows(-U)
    for _ in range(iters):
        tmp = -U + np.float32(pos_w) * kg.apply(Q) + np.float32(bi_w) * kb.apply(Q)
        Q = _softmax_rows(tmp)
    return Q.T.reshape(C, H, W)


class DenseCRF:
    """utils/dcrf.py:42-68."""

    def __init__(self, iter_max, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std):
        self.p = (iter_max, pos_w, pos_xy_std, bi_w, bi_xy_std, bi_rgb_std)

    def __call__(self, image, probmap):
        it, pw, ps, bw, bs, br = self.p
        return dense_crf_2d(np.ascontiguousarray(image), unary_from_softmax(probmap), it, pw, ps, bw, bs, br)


def crf_inference(img, probs, t=10, scale_factor=1, labels=21):
    """utils/dcrf.py:7-24."""
    return dense_crf_2d(np.ascontiguousarray(img), unary_from_softmax(probs), t, 3, 3 / scale_factor, 10, 80 / scale_factor, 13)


def crf_inference_label(img, labels, t=10, n_labels=21, gt_prob=0.7):
    """utils/dcrf.py:26-40."""
    U = unary_from_labels(labels, n_labels, gt_prob=gt_prob, zero_unsure=False)
    return np.argmax(dense_crf_2d(np.ascontiguousarray(img), U, t, 3, 3, 10, 50, 5), axis=0)
