"""Dev: error of the bf16x3 GEMM against a float64 product (max and rms, relative to the output's rms) for the library under EXCEL_AB_LIB,
and the tiny-net strip test's w_aff deviation: compares MFMA forms / accumulation orders.  python tools_dev/gemm_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
if os.environ.get("EXCEL_AB_LIB"):
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
tag = os.environ.get("EXCEL_AB_LIB", "shipped")
for (M, N, K) in ((4096, 768, 768), (4096, 768, 3072), (4096, 2304, 768)):
    rs = np.random.RandomState(K + N)
    A = rs.standard_normal((M, K)).astype(np.float32)
    W = (rs.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ W.T.astype(np.float64)
    out = ops.gemm_bf16x3(ops.split_bf16(torch.from_numpy(A).cuda()), ops.split_bf16(torch.from_numpy(W).cuda())).cpu().numpy().astype(np.float64)
    f32 = (torch.from_numpy(A).cuda().double() @ torch.from_numpy(W).cuda().double().T).float().cpu().numpy().astype(np.float64)   # fp64 product rounded to fp32: the floor
    e = out - ref
    s = np.sqrt((ref ** 2).mean())
    print(f"{tag} M={M} N={N} K={K}: max |err| / rms(out) = {np.abs(e).max() / s:.3e}, rms err / rms(out) = {np.sqrt((e**2).mean()) / s:.3e}   (fp32 rounding of the exact product: max {np.abs(f32 - ref).max() / s:.3e})")
