"""Dev probe (energy lens): the bf16x3 GEMM's time as a function of how many mantissa bits the LO planes carry.  The matrix pipe is power-capped by
its operand data (EXPERIMENTS.md round 4: 2 474 TFLOP/s on zero operands, 1 680-1 810 on random mantissas); the lo planes are pure mantissa noise.
Masks the low mantissa bits of both operands' lo planes on the host copy and times the unchanged kernel.  NOT a shipped mode: it changes the
numeric contract (16 -> 9 + kept bits of significand)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from excel_amd import ops
from excel_amd._lib import lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25120, 2304, 768)
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) * 0.05
ref = None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.empty((M, N), dtype=torch.float32, device="cuda")
for keep in (7, 5, 4, 3, 2, 0, -1):
    As, Ws = ops.split_bf16(A).clone(), ops.split_bf16(W).clone()
    for T in (As, Ws):
        flat = T.view(torch.int16).view(-1)
        idx = torch.arange(flat.numel(), device="cuda")
        lo = ((idx // 32) % 2) == 1                       # blocked layout: every 32 k = 32 hi then 32 lo (common.h split_off)
        if keep < 0: flat[lo] = 0
        elif keep < 7: flat[lo] = flat[lo] & torch.tensor(-(1 << (7 - keep)), dtype=torch.int16, device="cuda")
    f = lambda: lib().excel_gemm_bf16x3(As.data_ptr(), Ws.data_ptr(), out.data_ptr(), None, None, M, N, K, 0, 0, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    if ref is None: ref = out.clone()
    err = float((out - ref).abs().max() / ref.abs().max())
    print("lo planes keep %2d mantissa bits%s: %.1f us  %.1f TFLOP/s   max deviation from the full planes %.2e of the output range" % (
        max(keep, 0), " (lo = 0: plain bf16)" if keep < 0 else "", us, 2.0 * M * N * K / us / 1e6, err))
