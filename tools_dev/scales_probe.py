"""dev: forward time per input size of the multi-scale LAM fuse (batch 32 = 16 images + flips), COCO-shaped model"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from excel_amd.model import ExCEL_model
from excel_amd.tools import synthetic
from excel_amd._lib import lib
dev = torch.device("cuda", 0)
sd = synthetic.make_vit_state_dict(seed=0)
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=81, img_size=512, mode="train", device=dev, state_dict=sd, dataset_name="ms_coco",
                    num_atrr_clusters=224, text_features=synthetic.make_text_features(103))
for S in (256, 384, 512, 768):
    x = torch.randn(32, 3, S, S, device=dev)
    for _ in range(2): model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"S={S} N={(S//16)**2+1}: {dt*1e3:.1f} ms per 32 images")
