"""dev: harness_ragged with different decode thread counts (bench.harness_ragged)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from excel_amd.model import ExCEL_model
from excel_amd.tools import synthetic
dev = torch.device("cuda", 0)
sd = synthetic.make_vit_state_dict(seed=0)
model = ExCEL_model(clip_model="ExCEL_ViT-B/16", num_classes=21, img_size=448, mode="train", device=dev, state_dict=sd,
                    text_features=synthetic.make_text_features(45))
for w, si in ((16, 1e-3), (16, 5e-4), (16, 2.5e-4), (12, 5e-4), (24, 5e-4)):
    sys.setswitchinterval(si)
    r = bench.harness_ragged(model, dev, n_images=256, workers=w)
    print(w, si, r["images_per_s_resident"], r["images_per_s_with_decode"], r["images_per_s_with_decode_steady"], r["first_batch_s"])
