"""Dev experiment: one B=32 batch on one stream vs two B=16 half batches on two streams (tails of one fill under the other)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from excel_amd import ops
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
from excel_amd.tools import synthetic
dev = torch.device("cuda", 0)
B, S, NC = 32, 448, 21
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 8
sd, tf = synthetic.make_vit_state_dict(seed=0), synthetic.make_text_features(45)
ds = synthetic.SyntheticSegDataset(B, (S, S), num_classes=NC, seed=1234)
_, imgs, gts, cls = ds.batch(list(range(B)))
imgs, gts, cls = (torch.from_numpy(a).to(dev) for a in (imgs, gts, cls))
def mk():
    m = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=NC, img_size=S, mode="train", device=dev, state_dict=sd, text_features=tf)
    return TrainingFreePipeline(m, num_classes=NC, smax=ds.max_k())
pipes = [mk() for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
h = B // nsplit
parts = [(imgs[i*h:(i+1)*h].contiguous(), cls[i*h:(i+1)*h].contiguous(), gts[i*h:(i+1)*h].contiguous()) for i in range(nsplit)]
def step():
    for p, s, a in zip(pipes, streams, parts):
        with torch.cuda.stream(s):
            p.run_batch(*a)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"nsplit={nsplit}: {dt*1e3:.2f} ms/step  {B/dt:.1f} img/s")
t0 = time.perf_counter()
for _ in range(steps): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"   host enqueue time {(t1-t0)/steps*1e3:.2f} ms/step")
