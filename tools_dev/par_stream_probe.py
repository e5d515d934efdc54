"""dev: PAR at the benchmark's size (B = 32, 448^2, mean C ~ 3.5): the shipped recompute kernel vs the fp32 plane-streaming path
(EXCEL_PAR_STREAM_AFFINITIES) - what the chip sustains when the 48 weights are STREAMED (192 B/px of weights + 2C x 4) bounds what a
16-bit weight stream (96 B/px + 2C x 4) could reach before building it.   python tools_dev/par_stream_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from excel_amd import ops
B, H, W, Cmax = 32, 448, 448, 7
g = torch.Generator(device="cuda").manual_seed(0)
img = torch.randn(B, 3, H, W, device="cuda", generator=g)
masks = torch.rand(B, Cmax, H, W, device="cuda", generator=g)
nch = torch.tensor([3, 4, 3, 4] * 8, dtype=torch.int32, device="cuda")      # mean 3.5 channels (k + 1)
def run(stream):
    ops.prof_collect()
    for _ in range(2): ops.par_forward(img, masks, nchan=nch, stream_affinities=stream)
    torch.cuda.synchronize(); ops.prof_collect()
    ops.prof_enable(True, every=1)
    for _ in range(3): ops.par_forward(img, masks, nchan=nch, stream_affinities=stream)
    torch.cuda.synchronize(); ops.prof_enable(False)
    p = ops.prof_collect()
    return {k: (round(v["ms"] / 3, 4), v["launches"] // 3) for k, v in p.items() if v["launches"]}
px = B * H * W
for stream in (False, True):
    r = run(stream)
    it = r["par_iterate"]
    per = it[0] / it[1] * 1e3
    bytes_px = (48 * 4 + 2 * 3.5 * 4) if stream else ((5 + 3) * 4 + 2 * 3.5 * 4)
    print(f"stream={stream}: {r}  par_iterate {per:.1f} us/launch, moves {bytes_px:.0f} B/px -> {px * bytes_px / per / 1e6:.2f} TB/s", flush=True)
