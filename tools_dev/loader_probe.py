"""Dev: where does the ragged harness lose time?  loader alone (pinned / not), + H2D, + plan, + pipeline."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from excel_amd.tools import synthetic
from excel_amd.datasets import voc
from excel_amd.datasets.loader import ragged_batches
from excel_amd import ops
tmp = tempfile.mkdtemp(prefix="probe_")
root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
synthetic.write_voc_tree(root, lists, 256)
ds = voc.VOC12SegDataset(root_dir=root, name_list_dir=lists, split="train", stage="val")
order = list(range(256)) * 4
torch.zeros(1).cuda()
t0 = time.perf_counter(); x = [ds[i] for i in range(32)]; print("decode ms/img in-process", (time.perf_counter() - t0) / 32 * 1e3)
for workers, pin in ((16, False), (16, True), (32, True)):
    t0 = time.perf_counter(); tf = None; n = 0
    for rb in ragged_batches(ds, order, 32, num_workers=workers, pin_memory=pin):
        if tf is None: tf = time.perf_counter()
        n += len(rb)
    t1 = time.perf_counter()
    print(f"loader only workers={workers} pin={pin}: first batch {tf - t0:.2f}s, steady {(n - 32) / (t1 - tf):.0f} img/s")
dev = torch.device("cuda")
t0 = time.perf_counter(); tf = None; n = 0; tp = 0.0
for rb in ragged_batches(ds, order, 32, num_workers=16):
    if tf is None: tf = time.perf_counter()
    a = time.perf_counter()
    im = rb.images.to(dev, non_blocking=True); lb = rb.labels.to(dev, non_blocking=True); c = rb.cls.to(dev, non_blocking=True)
    plan = ops.RaggedPlan(rb.hw, dev)
    tp += time.perf_counter() - a
    n += len(rb)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"loader + H2D + plan: steady {(n - 32) / (t1 - tf):.0f} img/s; main-thread H2D+plan {tp / (n / 32) * 1e3:.2f} ms/batch")
shutil.rmtree(tmp, ignore_errors=True)

# ---- with the pipeline
import bench as _b
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
tmp = tempfile.mkdtemp(prefix="probe_")
root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
synthetic.write_voc_tree(root, lists, 256)
ds = voc.VOC12SegDataset(root_dir=root, name_list_dir=lists, split="train", stage="val")
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=448, mode="train", device=dev,
                    state_dict=synthetic.make_vit_state_dict(seed=0), text_features=synthetic.make_text_features(45))
pipe = TrainingFreePipeline(model, num_classes=21, smax=6)
x = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); t0 = time.perf_counter(); y = x.to(dev, non_blocking=True); torch.cuda.synchronize(); print("H2D GB/s pinned 64MB", 0.064 / (time.perf_counter() - t0))
from excel_amd.datasets.loader import DeviceFeeder
for workers in (16, 16, 8, 32):
    t0 = time.perf_counter(); tf = None; n = 0
    for names, plan, im, c, lb in DeviceFeeder(ragged_batches(ds, order, 32, num_workers=workers, pin_memory=False), dev):
        if tf is None: tf = time.perf_counter()
        pipe.run_batch_ragged(im, plan, c, lb)
        n += len(names)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"DeviceFeeder workers={workers}: first batch {tf - t0:.2f}s, steady {(n - 32) / (t1 - tf):.0f} img/s, total {n / (t1 - t0):.0f} img/s")
shutil.rmtree(tmp, ignore_errors=True)
