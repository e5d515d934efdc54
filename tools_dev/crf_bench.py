"""Dev: DenseCRF stage per image at a VOC size (375x500, 3 channels, tools/infer_lam.py:191-198 parameters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from excel_amd import ops
rs = np.random.RandomState(0)
H, W, C = 375, 500, 3
img = torch.from_numpy((rs.rand(H, W, 3) * 255).astype(np.uint8)).cuda()
p = rs.rand(C, H, W).astype(np.float32); p /= p.sum(0)
p = torch.from_numpy(p).cuda()
for _ in range(3): ops.dcrf_inference(img, p, 10, 3, 1, 4, 67, 3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.dcrf_inference(img, p, 10, 3, 1, 4, 67, 3)
torch.cuda.synchronize(); print("dcrf 375x500x3, 10 iterations: %.2f ms per image" % ((time.perf_counter() - t0) / 20 * 1e3))
