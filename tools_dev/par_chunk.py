"""Dev experiment: PAR over the whole batch vs over image chunks whose aff planes fit the 256 MB Infinity Cache."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from excel_amd import ops
B, C, S = 32, 3, 448
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randn(B, 3, S, S, device="cuda", generator=g)
masks = torch.rand(B, C, S, S, device="cuda", generator=g)
nchan = torch.full((B,), 2, dtype=torch.int32, device="cuda"); nchan[::3] = 3
def run(chunk):
    outs = []
    for s in range(0, B, chunk):
        outs.append(ops.par_forward(imgs[s:s+chunk], masks[s:s+chunk], nchan=nchan[s:s+chunk]))
    return outs
for chunk in (32, 16, 8, 6, 4, 2):
    run(chunk); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run(chunk)
    torch.cuda.synchronize()
    print(f"chunk {chunk:2d}: {(time.perf_counter()-t0)/5*1e3:.2f} ms per batch of {B}")
