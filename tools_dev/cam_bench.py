"""Dev: the patch-text CAM kernels alone at the production shape (B=32, N=785, C=512, T=45), for a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -o cam -- python tools_dev/cam_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from excel_amd import ops
B, N, C, T, F = int(os.environ.get("B", 32)), 785, 512, 45, 20
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.standard_normal((B, N, C)).astype(np.float32)).cuda()
t = rs.standard_normal((T, C)).astype(np.float32)
t = torch.from_numpy(t / np.linalg.norm(t, axis=1, keepdims=True)).cuda()
for mode in ("bf16x3",):
    for _ in range(5):
        ops.patch_text_cam(x, t, num_fg=F, mode=mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.patch_text_cam(x, t, num_fg=F, mode=mode)
    e1.record()
    torch.cuda.synchronize()
    print(mode, "ms per call (incl. python + text split + 3 kernels):", e0.elapsed_time(e1) / 50)
