"""Dev (EXCEL_DEV=1 build): per-shape times of the bf16x3 GEMM and of its ablation arms -> JSON (profiles/r03_gemm_ceiling.json).
   arms: full | no_dma (EXCEL_BF_DBG=2: staging loads only for the first tile: MFMAs + fragment reads + barriers on stale LDS)
         | mfma_only (EXCEL_BF_DBG=4: the kernel's MFMAs back to back, operands in registers: the power-limited ceiling)
   Every arm is a separate process (the dev library reads EXCEL_BF_DBG once).  usage: python tools_dev/gemm_ceiling.py > out.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("qkv", 25120, 2304, 768), ("proj", 25120, 768, 768), ("fc1", 25120, 3072, 768), ("fc2", 25120, 768, 3072), ("cam_proj", 25120, 512, 768)]
CHILD = r'''
import sys, os, json, torch, ctypes as C
sys.path.insert(0, %r)
from excel_amd import ops
from excel_amd._lib import lib
out = {}
for name, M, N, K in %r:
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    As, Ws = ops.split_bf16(A), ops.split_bf16(W)
    o = torch.empty((M, N), dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: lib().excel_gemm_bf16x3(As.data_ptr(), Ws.data_ptr(), o.data_ptr(), None, None, M, N, K, 0, 0, st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    out[name] = {"M": M, "N": N, "K": K, "us": round(us, 1), "tflops_fp32_equiv": round(2.0 * M * N * K / us / 1e6, 1)}
print(json.dumps(out))
''' % (ROOT, SHAPES)
res = {}
for arm, dbg in (("full", "0"), ("no_dma", "2"), ("mfma_only", "4")):
    env = dict(os.environ, EXCEL_BF_DBG=dbg)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr); sys.exit(1)
    res[arm] = json.loads(p.stdout.strip().splitlines()[-1])
table = {}
for name, M, N, K in SHAPES:
    full, nod, mf = (res[a][name] for a in ("full", "no_dma", "mfma_only"))
    # 3 bf16 MFMAs per fp32 product: issue rate of the MFMA-only arm as a fraction of the dense bf16 peak
    table[name] = {"M": M, "N": N, "K": K, "full_us": full["us"], "no_dma_us": nod["us"], "mfma_only_us": mf["us"],
                   "full_tflops_fp32_equiv": full["tflops_fp32_equiv"], "frac_of_2500": round(full["tflops_fp32_equiv"] / 2500, 4),
                   "ceiling_frac_of_2500": round(mf["tflops_fp32_equiv"] / 2500, 4), "full_over_ceiling": round(mf["us"] / full["us"], 3)}
print(json.dumps({"_note": "bf16x3 GEMM (3 x v_mfma_f32_32x32x16_bf16 per fp32 product): full kernel vs ablation arms of the SAME kernel "
                           "(dev build, tools_dev/gemm_ceiling.py). mfma_only = the kernel's MFMA stream with operands in registers, no LDS / DMA / "
                           "barriers: what the part sustains on this instruction mix (clock / power limited); frac_of_2500 = fp32-equivalent "
                           "TFLOP/s / 2500 (bench.py roofline.frac of the shape)", "shapes": table}, indent=1))
