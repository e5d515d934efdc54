"""Dev tool: single-GPU timings of the BASELINE configs that are not the bench line (numbers quoted in DESIGN.md section 4).
    configs[1]  VOC 448x448 batch=16: ViT attention + patch-text CAM only (no affinity / PAR)
    configs[4]  COCO-shaped 512x512 batch=16 per GPU: 80-class path (T=103 text rows, 224-cluster bank) incl. PAR, and the
                flip + multi-scale LAM fuse on its own
Usage: python tools_dev/configs_bench.py [steps] [fp16]        (fp16: checkpoint-like weights - rounded through IEEE half like the published CLIP
archive; the model then starts in f16x2)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
from excel_amd.tools import synthetic
from excel_amd.utils.camutils import multi_scale_lam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
sd = synthetic.make_vit_state_dict(seed=0)
if len(sys.argv) > 2 and sys.argv[2] == "fp16":
    import numpy as np
    sd = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---- configs[1]
B, S = 16, 448
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=S, mode="train", device=dev, state_dict=sd,
                    text_features=synthetic.make_text_features(45))
ds = synthetic.SyntheticSegDataset(B, (S, S), num_classes=21, seed=7)
_, imgs, gts, cls = ds.batch(range(B))
x = torch.from_numpy(imgs).to(dev)
dt = timed(lambda: model(x), steps)
print("gemm mode:", model.encoder.visual.handle().gemm_mode())
print(f"configs[1] VOC 448^2 batch=16, ViT + patch-text CAM only: {dt * 1e3:.2f} ms/step  {B / dt:.1f} images/s")
del model

# ---- configs[4]
B, S = 16, 512
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=81, img_size=S, mode="train", device=dev, state_dict=sd, dataset_name="ms_coco",
                    num_atrr_clusters=224, text_features=synthetic.make_text_features(103))
ds = synthetic.SyntheticSegDataset(B, (S, S), num_classes=81, seed=99)
_, imgs, gts, cls = ds.batch(range(B))
bx = (torch.from_numpy(imgs).to(dev), torch.from_numpy(cls).to(dev), torch.from_numpy(gts).to(dev))
pipe = TrainingFreePipeline(model, num_classes=81, smax=ds.max_k(), caa_thre=0.88)


def step():
    pipe.run_batch(*bx)
    pipe.drain()


dt = timed(step, steps)
print(f"configs[4] COCO-shaped 512^2 batch=16 (N=1025, T=103, F=80), full path incl. PAR: {dt * 1e3:.2f} ms/step  {B / dt:.1f} images/s")
dt = timed(lambda: multi_scale_lam(model, bx[0], scales=(1.0, 0.5, 0.75, 1.5)), max(2, steps // 3))
print(f"configs[4] flip + multi-scale LAM fuse (scales 1, 0.5, 0.75, 1.5) batch=16: {dt * 1e3:.2f} ms/batch  {B / dt:.1f} images/s")
