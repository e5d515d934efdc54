"""Dev probe: the same bf16x3 GEMMs on two streams at once (plus LDS-using neighbours on a third) against their serial results, bit for bit.
python tools_dev/w4_concurrency_probe.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("EXCEL_AB_LIB"):
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (12560, 768, 768)
NG = int(os.environ.get("PROBE_GEMM_STREAMS", "2")); LN = int(os.environ.get("PROBE_LN", "1")); REPS = int(os.environ.get("PROBE_REPS", "3"))
g = torch.Generator(device="cuda").manual_seed(0)
As = [ops.split_bf16(torch.randn(M, K, device="cuda", generator=g)) for _ in range(2)]
W = ops.split_bf16(torch.randn(N, K, device="cuda", generator=g) * 0.05)
bias = torch.randn(N, device="cuda", generator=g)
ref = [ops.gemm_bf16x3(a, W, bias=bias) for a in As]
refs = [ops.gemm_bf16x3(a, W, bias=bias, split_out=True).clone() for a in As]
torch.cuda.synchronize()
x = torch.randn(25120, 768, device="cuda", generator=g)
lw, lb = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
streams = [torch.cuda.Stream() for _ in range(3)]
bad = 0
for rep in range(REPS):
    outs = [[], []]
    for i in range(NG):
        with torch.cuda.stream(streams[i]):
            for _ in range(8):
                outs[i].append(ops.gemm_bf16x3(As[i], W, bias=bias))
                outs[i].append(ops.gemm_bf16x3(As[i], W, bias=bias, split_out=True))
    with torch.cuda.stream(streams[2]):
        for _ in range(20 if LN else 0):
            y = ops.layernorm(x, lw, lb)
    torch.cuda.synchronize()
    print('rep', rep, 'done', flush=True)
    for i in range(NG):
        for j, o in enumerate(outs[i]):
            r = refs[i] if j % 2 else ref[i]
            if not torch.equal(o, r):
                bad += 1
                if bad < 4:
                    d = (o.float() - r.float()).abs() if o.dtype == torch.float32 else (o != r)
                    print("mismatch rep", rep, "stream", i, "launch", j, "count", int((o != r).sum()), "first rows", torch.nonzero((o != r).reshape(M, -1).any(1))[:8].flatten().tolist())
print("shape", M, N, K, "mismatching launches:", bad, "of", 10 * 2 * 16)
