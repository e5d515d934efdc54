"""Dev: whole-path throughput at other shapes (spot pathological tile choices)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from excel_amd import ops
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
from excel_amd.tools import synthetic
sd = synthetic.make_vit_state_dict(seed=0)
for (B, S, nc, T, ds_name, K) in ((32, 448, 21, 45, "pascal_voc", 112), (16, 448, 21, 45, "pascal_voc", 112), (32, 320, 21, 45, "pascal_voc", 112),
                                  (16, 512, 81, 103, "ms_coco", 224), (64, 448, 21, 45, "pascal_voc", 112)):
    model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=nc, img_size=S, mode="train", state_dict=sd, dataset_name=ds_name,
                        num_atrr_clusters=K, text_features=synthetic.make_text_features(T))
    ds = synthetic.SyntheticSegDataset(B, (S, S), num_classes=nc, seed=1234)
    _, imgs, gts, cls = ds.batch(range(B))
    imgs, gts, cls = (torch.from_numpy(a).cuda() for a in (imgs, gts, cls))
    pipe = TrainingFreePipeline(model, num_classes=nc, smax=ds.max_k())
    for _ in range(2): pipe.run_batch(imgs, cls, gts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): pipe.run_batch(imgs, cls, gts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ops.prof_collect(); ops.prof_enable(True, every=1)
    pipe.run_batch(imgs, cls, gts); torch.cuda.synchronize(); ops.prof_enable(False)
    pr = ops.prof_collect()
    top = sorted(((k, v["ms"]) for k, v in pr.items() if v["launches"]), key=lambda kv: -kv[1])[:5]
    print(f"B={B} S={S} nc={nc}: {B/dt:7.1f} img/s  {dt*1e3:6.2f} ms/step  per-image {dt/B*1e3:.3f} ms   top: " + ", ".join(f"{k} {m:.2f}" for k, m in top))
    del model, pipe
