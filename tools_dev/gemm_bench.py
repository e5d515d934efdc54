"""Standalone GEMM micro-benchmark (dev tool): python tools_dev/gemm_bench.py [M N K] [reps] [mode]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("EXCEL_AB_LIB"):      # A/B a differently built library on the same box
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25120, 3072, 768)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
mode = sys.argv[5] if len(sys.argv) > 5 else "bf16x3"
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * 0.05
if mode.startswith("bf16x3"):
    import ctypes as C
    from excel_amd._lib import lib
    As, Ws = ops.split_bf16(A), ops.split_bf16(W)
    out = torch.empty((M, 2 * N if mode.endswith("split") else N), dtype=torch.float32, device="cuda")
    so = {"bf16x3": 0, "bf16x3_split": 1, "bf16x3_noepi": 99, "bf16x3_res": 0, "bf16x3_gelu_split": 1}[mode]
    act = 1 if mode == "bf16x3_gelu_split" else 0          # bias + QuickGELU + split output: the fc1 form
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = torch.randn(M, N, device="cuda", generator=g) if mode == "bf16x3_res" else None      # bias + residual epilogue (out-proj / fc2 form)
    bias = torch.randn(N, device="cuda", generator=g) if mode in ("bf16x3_res", "bf16x3_gelu_split") else None
    f = lambda: lib().excel_gemm_bf16x3(As.data_ptr(), Ws.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        res.data_ptr() if res is not None else None, M, N, K, act, so, st)
else:
    f = lambda: ops.gemm(A, W)
for _ in range(3): f()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps): f()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
print(f"{mode} M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s (fp32-equivalent)")
