import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get('NT'): torch.set_num_threads(int(os.environ['NT']))
from excel_amd.tools import synthetic
from excel_amd.datasets import voc
from excel_amd.datasets.loader import ragged_batches, threaded_batches, DeviceFeeder
from excel_amd import ops
tmp = tempfile.mkdtemp(prefix="probe_")
root, lists = os.path.join(tmp, "VOC2012"), os.path.join(tmp, "lists")
synthetic.write_voc_tree(root, lists, 256)
ds = voc.VOC12SegDataset(root_dir=root, name_list_dir=lists, split="train", stage="val")
order = list(range(256)) * 2
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
# 0. instrumented copy of the feeder loop with the pipeline
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=448, mode="train", device=dev,
                    state_dict=synthetic.make_vit_state_dict(seed=0), text_features=synthetic.make_text_features(45))
pipe = TrainingFreePipeline(model, num_classes=21, smax=6)
import threading, queue
Probe = DeviceFeeder
for it_ in range(3):
    fd = Probe(threaded_batches(ds, order, 32, num_threads=int(os.environ.get("TH", 16))), dev)
    t0 = time.perf_counter(); n = 0; tr = 0.0; evs = []
    for names, plan, im, c, lb in fd:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a = time.perf_counter(); e0.record(); pipe.run_batch_ragged(im, plan, c, lb); e1.record(); tr += time.perf_counter() - a; n += len(names); evs.append((e0, e1))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g = [a.elapsed_time(b) for a, b in evs]
    print(f"with pipeline: {n / dt:.0f} img/s; consumer run_batch {tr / len(evs) * 1e3:.1f} ms/batch; GPU ms/batch mean {np.mean(g):.1f} max {np.max(g):.1f}")
shutil.rmtree(tmp, ignore_errors=True)
