"""Dev (EXCEL_DEV library, EXCEL_W4_DBG=136 or 143): cycle stamps of the four-wave GEMM's k-loop - s_memtime before / after every step's
barrier on wave 0 of workgroup 0 -> cycles per 32-k step and cycles spent in the barrier."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("EXCEL_AB_LIB"):
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
from excel_amd._lib import lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25120, 2304, 768)
ACT = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: QuickGELU (the fc1 form)
X2 = len(sys.argv) > 5 and sys.argv[5] == "x2"              # the two-product compact-weight kernel (gemm_w4x2.hip) on fp16-valued weights
NMFMA = 160 if X2 else 240
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) * 0.05
if X2:
    W = W.half().float()
    As, Ws = ops.split_bf16(A, f16=True), ops.split_bf16(W, f16=True)
    Wh, _bad = ops.pack_f16(W)
else:
    As, Ws = ops.split_bf16(A), ops.split_bf16(W)
out = torch.empty((M, 2 * N), dtype=torch.float32, device="cuda")
stamps = torch.zeros(max(N, 8192), dtype=torch.float32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    if X2: lib().excel_gemm_f16x2(As.data_ptr(), Ws.data_ptr(), Wh.data_ptr(), out.data_ptr(), stamps.data_ptr(), None, M, N, K, ACT, 1, st)
    else: lib().excel_gemm_bf16x3(As.data_ptr(), Ws.data_ptr(), out.data_ptr(), stamps.data_ptr(), None, M, N, K, ACT, 1, st)
torch.cuda.synchronize()
t = stamps.view(torch.int64)[: 2 * (K // 32)].cpu().numpy().reshape(-1, 2)
rt = stamps.view(torch.int64)[256: 256 + K // 32].cpu().numpy()
step = t[1:, 0] - t[:-1, 0]
print("  shader clock over the k-loop: %d cycles in %d ticks of the 100 MHz real-time counter -> %.2f GHz" % (t[-1, 0] - t[0, 0], rt[-1] - rt[0], (t[-1, 0] - t[0, 0]) / max(rt[-1] - rt[0], 1) * 0.1))
print("dbg", os.environ.get("EXCEL_W4_DBG"), "shape", M, N, K, "| cycles per step (barrier arrival to next barrier arrival):", step.tolist())
print("  mean step %.0f (" + str(NMFMA) + " MFMAs x 16 = " + str(NMFMA * 16) + "), barrier wait per step mean %.0f, max %.0f; s_memtime ticks are at 100 MHz x? check: total %.0f ticks" % (step.mean(), (t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).max(), t[-1, 0] - t[0, 0]))
ph = stamps.view(torch.int64)[400:424].cpu().numpy().reshape(3, 8)[:, :4]
for wgi, r in enumerate(ph):
    if r[0]:
        print("  workgroup %3d phases (us): prologue %.2f  k-loop %.2f  drain+epilogue %.2f  | entry at +%.2f us after workgroup 0's" % (
            100 * wgi, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100 if r[3] else float("nan"), (r[0] - ph[0][0]) / 100))
import numpy as np
nt = min(-(-M // 320) * -(-N // 256), 700)
se = stamps.view(torch.int64)[500:500 + 2 * nt].cpu().numpy().reshape(nt, 2).astype(np.float64) / 100.0
t0 = se[:, 0].min()
dur = se[:, 1] - se[:, 0]
print("  all %d workgroups: entry spread %.2f us, duration min / median / max %.1f / %.1f / %.1f us, last end at %.1f us; slowest: %s" % (
    nt, se[:, 0].max() - t0, dur.min(), np.median(dur), dur.max(), se[:, 1].max() - t0, [(int(i), round(float(dur[i]), 1)) for i in np.argsort(-dur)[:6]]))
print("  duration by XCD (b % 8):", [round(float(np.median(dur[x::8])), 1) for x in range(8)], " second-round entries:", int((se[:, 0] - t0 > 20).sum()))

ep = stamps.view(torch.int64)[2000:2013].cpu().numpy().astype(np.float64)
if ep[0]:
    print("  epilogue of workgroup 0, wave 0 (cycles): entry -> bias loads issued %.0f | per row tile (stores issued): %s | total %.0f" % (
        ep[1] - ep[0], [int(ep[2 + i] - ep[1 + i]) for i in range(10)], ep[11] - ep[0]))
first = se[:, 0] - t0 < 20.0
if (~first).any():
    print("  first-round workgroups: median duration %.1f us; later rounds: %.1f us (an epilogue that misses the instruction cache only in the first round shows here)" % (
        float(np.median(dur[first])), float(np.median(dur[~first]))))
