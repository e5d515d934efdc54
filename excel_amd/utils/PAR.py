"""PAR(dilations, num_iter)(imgs, masks) - mirror of utils/PAR.py:26-92 over the HIP kernels."""
import torch

from .. import ops


class PAR:
    def __init__(self, dilations, num_iter):
        self.dilations = list(dilations)
        self.num_iter = num_iter
        self.w1 = 0.3      # PAR.py:35
        self.w2 = 0.01     # PAR.py:36

    def cuda(self, *a, **k):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def forward(self, imgs, masks, nchan=None):
        """imgs [B,3,h,w], masks [B,C,H,W] -> refined masks [B,C,H,W] (PAR.py:64-92)."""
        return ops.par_forward(imgs, masks, self.dilations, self.num_iter, nchan=nchan, w1=self.w1, w2=self.w2)

    __call__ = forward
