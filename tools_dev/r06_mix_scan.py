"""dev (EXCEL_DEV library built with tools_dev/w4_mixspec.patch applied - the EXCEL_W4_MIXSPEC knob lives in that patch, not in the tree -, saved as tools_dev/ab/dev.so): scan the split of a two-instance GEMM launch - `tall` row tiles of 320 rows + the rest in
256- / 160-row tiles - per layer shape, against the uniform launch (EXCEL_W4_MIX=0) and the launcher's own pick.  One process per point
(the knob is read once).   python tools_dev/r06_mix_scan.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import torch, excel_amd._lib as _L
_L.LIB_PATH = os.path.abspath("tools_dev/ab/dev.so")
from excel_amd import ops
M, N, K, x2 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "x2"
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g); W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half().float()
bias = torch.randn(N, device="cuda", generator=g)
As, Ws = ops.split_bf16(A, f16=x2), ops.split_bf16(W, f16=x2)
Wh = ops.pack_f16(W)[0] if x2 else None
f = (lambda: ops.gemm_f16x2(As, Ws, Wh, bias=bias, split_out=True)) if x2 else (lambda: ops.gemm_bf16x3(As, Ws, bias=bias, split_out=True))
for _ in range(5): f()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
print("%%.1f" %% best)
''' % ROOT
def run(M, N, K, kind, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, str(M), str(N), str(K), kind], capture_output=True, text=True, env=e, cwd=ROOT)
    try:
        return float(r.stdout.strip().splitlines()[-1])
    except Exception:
        return float("nan")
SHAPES = [(25120, 2304, 768, range(44, 72, 4)), (25120, 3072, 768, range(40, 76, 4)), (12560, 2304, 768, range(8, 36, 4)), (16400, 3072, 768, range(16, 48, 4)),
          (16400, 2304, 768, range(16, 48, 4)), (12560, 3072, 768, range(8, 36, 4))]
for M, N, K, talls in SHAPES:
    for kind in ("x3", "x2"):
        uni = run(M, N, K, kind, {"EXCEL_W4_MIX": "0"})
        pick = run(M, N, K, kind, {})
        pts = []
        for sec in (8, 5):
            for t in talls:
                pts.append((run(M, N, K, kind, {"EXCEL_W4_MIXSPEC": "%d:%d" % (t, sec)}), t, sec))
        pts.sort()
        print(f"{M}x{N}x{K} {kind}: uniform {uni:.1f} us, launcher's pick {pick:.1f}; best splits (us, tall, second): {[(round(a, 1), b, c) for a, b, c in pts[:4]]}", flush=True)
