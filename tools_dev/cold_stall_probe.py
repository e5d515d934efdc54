"""Dev probe: which action makes a later bench step stall 40-60 ms on the host when kernels are bracketed with timing events?
python tools_dev/cold_stall_probe.py <variant>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from excel_amd import ops
class A: batch = 32
dev = torch.device("cuda", 0)
pipe, batches, ks, model = bench.make_workload(A, 0, 1, dev)
variant = sys.argv[1] if len(sys.argv) > 1 else "d"
cats = ["gemm_bf16x3", "par_iterate"]
def loop(n, tag, sync_each=False):
    hs = []
    t0 = time.perf_counter()
    for i in range(n):
        t = time.perf_counter(); pipe.run_batch(*batches[i % 2]); hs.append(round((time.perf_counter() - t) * 1e3, 2))
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{variant} {tag}: {hs}  total {1e3*(time.perf_counter()-t0)/n:.2f} ms/step", flush=True)
ops.prof_enable(True, categories=cats, every=4)
loop(2, "warm")
if variant == "d":
    ops.prof_collect(); loop(10, "after collect")
elif variant == "e":
    loop(10, "no collect")
elif variant == "f":
    ops.prof_collect(); torch.cuda.synchronize(); time.sleep(0.5); loop(10, "after collect + sleep")
elif variant == "g":
    pipe.reset(); loop(10, "after pipe.reset, no collect")
elif variant == "h":
    ops.prof_collect(); loop(3, "after collect, sync each", sync_each=True); loop(10, "then free-running")
elif variant == "i":
    ops.prof_enable(True, categories=cats, every=1); loop(10, "no collect, every=1")
