"""Dev probe: socket power while ONE stage of the benchmark step runs in a loop (rocm-smi sampled from a side thread), and the
stage's time -> joules per step per stage.  Stages: vit (model forward: GEMMs + attention + LayerNorm + CAM), par (PAR step
iterations + statistics), gemm_qkv / gemm_fc2 (one layer shape of the bf16x3 GEMM), full (the whole step).
python tools_dev/stage_power.py [seconds per stage]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from excel_amd import ops
class A: batch = 32
dev = torch.device("cuda", 0)
pipe, batches, ks, model = bench.make_workload(A, 0, 1, dev)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
inputs, cls, gts = batches[0]
B, S = 32, 448
g = S // 16

def sampler(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            w = re.findall(r"Package Power \(W\): (\d+\.\d+)", t); c = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", t)
            if w and c: out.append((float(w[0]), int(c[0])))
        except Exception:
            pass
        time.sleep(0.2)

def run(tag, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(4): fn()
        torch.cuda.synchronize(); n += 4
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    mid = sorted(out[2:-1] if len(out) > 4 else out)       # drop the ramp at both ends
    w = sum(x[0] for x in mid) / max(len(mid), 1); c = sum(x[1] for x in mid) / max(len(mid), 1)
    print(f"{tag:10s} {dt*1e3:8.3f} ms per call   {w:7.0f} W  {c:6.0f} MHz   {w*dt:7.2f} J per call   ({len(mid)} samples)", flush=True)

# pieces of the step
_, _, attr, attn_w, _ = model(inputs)
idx, ncls, nchan = ops.cls_compact(cls, pipe.smax, want_nchan=True)
refined = ops.refine_cams_with_aff_batched(attr, attn_w.w_aff, idx, ncls, g, pipe.caa_thre)
C = pipe.smax + 1
cams = ops.cam_upsample_bkg(refined, ncls, g, S, S)
ws = torch.empty(ops.lib().excel_par_workspace_bytes(B, C, S, S, len(pipe.dilations)), dtype=torch.uint8, device=dev)
par_out = torch.empty_like(cams)
import ctypes as CT
from excel_amd._lib import lib
def gemm_fn(M, N, K, split):
    gen = torch.Generator(device="cuda").manual_seed(0)
    Am = torch.randn(M, K, device="cuda", generator=gen); W = torch.randn(N, K, device="cuda", generator=gen) * 0.05
    As, Ws = ops.split_bf16(Am), ops.split_bf16(W)
    out = torch.empty((M, 2 * N if split else N), dtype=torch.float32, device="cuda")
    st = CT.c_void_p(torch.cuda.current_stream().cuda_stream)
    return lambda: [lib().excel_gemm_bf16x3(As.data_ptr(), Ws.data_ptr(), out.data_ptr(), None, None, M, N, K, 0, 1 if split else 0, st) for _ in range(20)]
run("idle", lambda: time.sleep(0.05))
run("full step", lambda: pipe.run_batch(inputs, cls, gts))
run("vit+cam", lambda: model(inputs))
run("par", lambda: ops.par_forward(inputs, cams, pipe.dilations, pipe.num_iter, nchan=nchan, ws=ws, out=par_out))
run("gemm qkv x20", gemm_fn(25120, 2304, 768, True))
run("gemm fc2 x20", gemm_fn(25120, 768, 3072, False))
