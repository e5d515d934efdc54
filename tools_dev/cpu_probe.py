import numpy as np, time, sys, os
sys.path.insert(0, '.')
import oracle, torch
from oracle import torch_cpu
from oracle.vit import VitConfig
from excel_amd.tools import synthetic
cfg = VitConfig(width=768, layers=12, heads=12, patch=16, out_dim=512, input_resolution=224, n_surgery=5)
w = oracle.vit.reload_self_attn(synthetic.make_vit_state_dict(seed=0), cfg, 28, "train")
bank = np.load("excel_amd/attributes_text/attr_bank_pascal_voc.npz")["bank"]
text_attr = oracle.attr.attr_aggregate(synthetic.make_text_features(45), bank, 20)
ds = synthetic.SyntheticSegDataset(2, (448, 448), seed=1234)
samples = [(ds[i][1], ds[i][2], ds[i][3]) for i in range(2)]
print("cpus", os.cpu_count(), flush=True)
for th in (int(a) for a in sys.argv[1:]):
    t0 = time.time()
    h, p, st = torch_cpu.build_validation(samples, w, cfg, text_attr, 21, 448, threads=th)
    print(th, round(time.time() - t0, 2), {k: round(v, 2) for k, v in st.items()}, flush=True)
