"""dev: f16x3 vs f16x2 (split-layout weights / compact half weights) on the layer shapes, same process, interleaved.
python tools_dev/f16x2_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("EXCEL_AB_LIB"):      # A/B a differently built library on the same box
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
ONLY = os.environ.get("F16X2_ONLY")      # "half": time only the compact-weight two-product kernel (A/B runs)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [(25120, 768, 768, "proj+res"), (25120, 2304, 768, "qkv"), (25120, 3072, 768, "fc1"), (25120, 768, 3072, "fc2+res"),
          (25120, 512, 768, "final proj"), (12560, 768, 768, "b16 proj"), (12560, 2304, 768, "b16 qkv"), (12560, 3072, 768, "b16 fc1"),
          (12560, 768, 3072, "b16 fc2"), (16400, 2304, 768, "coco qkv"), (16400, 3072, 768, "coco fc1")]
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K, name in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half().float()       # fp16-valued, like a CLIP archive
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g) if "res" in name else None
    act = 1 if "fc1" in name else 0
    so = "fc1" in name
    As, Ws = ops.split_bf16(A, f16=True), ops.split_bf16(W, f16=True)
    Wh, bad = ops.pack_f16(W)
    assert bad == 0
    Ab, Wb = ops.split_bf16(A), ops.split_bf16(W)
    t = {}
    for rnd in range(2):
        if ONLY == "half":
            t.setdefault("f16x2/half", []).append(timeit(lambda: ops.gemm_f16x2(As, Ws, Wh, bias=bias, residual=res, act=act, split_out=so)))
            continue
        t.setdefault("bf16x3", []).append(timeit(lambda: ops.gemm_bf16x3(Ab, Wb, bias=bias, residual=res, act=act, split_out=so)))
        t.setdefault("f16x3", []).append(timeit(lambda: ops.gemm_bf16x3(As, Ws, bias=bias, residual=res, act=act, split_out=so, f16=True)))
        t.setdefault("f16x2/split", []).append(timeit(lambda: ops.gemm_f16x2(As, Ws, None, bias=bias, residual=res, act=act, split_out=so)))
        t.setdefault("f16x2/half", []).append(timeit(lambda: ops.gemm_f16x2(As, Ws, Wh, bias=bias, residual=res, act=act, split_out=so)))
    o3 = ops.gemm_bf16x3(As, Ws, bias=bias, residual=res, act=act, split_out=so, f16=True)
    same = [bool(torch.equal(o3, ops.gemm_f16x2(As, Ws, wh, bias=bias, residual=res, act=act, split_out=so))) for wh in (None, Wh)]
    print(f"{name:11s} M={M} N={N} K={K}: " + "  ".join(f"{k} {min(v):.1f} us" for k, v in t.items()) + f"  bit-identical {same}", flush=True)
