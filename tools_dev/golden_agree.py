"""Dev: label agreement / histogram distance of the golden end-to-end trace (tests/golden/pipeline_tiny.npz) in both GEMM modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.vit import VitConfig, make_vit_weights
from excel_amd import ops
from excel_amd.model import ExCEL_model
from excel_amd.utils import evaluate
from excel_amd.utils.affutils import refine_cams_with_aff, refine_cams_with_bkg_weclip
from excel_amd.utils.PAR import PAR
TINY = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)
KW = dict(width=128, layers=8, heads=2, patch=16, output_dim=64, input_resolution=64)
g = np.load("tests/golden/pipeline_tiny.npz")
for mode in ("f32", "bf16x3"):
    model = ExCEL_model(clip_model="tiny", num_classes=5, img_size=96, mode="train", state_dict=make_vit_weights(TINY, seed=11), vit_cfg=KW,
                        text_attr=g["text"].T.copy(), gemm_mode=mode)
    par = PAR(num_iter=20, dilations=[1, 2, 4, 8, 12, 24])
    gts, preds = [], []
    for i in range(4):
        inputs = ops.bilinear_resize(torch.from_numpy(g[f"s{i}_img"][None]).cuda(), 96, 96, align_corners=False)
        _, _, maps, aw, _ = model(inputs)
        refined, cls_lst = refine_cams_with_aff(maps[0], aw[:, 0], torch.from_numpy(g[f"s{i}_cls"]).cuda(), size=inputs.shape[2:], caa_thre=0.79)
        H, W = g[f"s{i}_gt"].shape
        labels, cams = refine_cams_with_bkg_weclip(refined, inputs[0], cls_lst, par, (H, W))
        lab = labels.cpu().numpy()[0]
        print(mode, i, "px", lab.size, "mismatch", int((lab != g[f"s{i}_label"]).sum()), "maps err", float(np.abs(maps.cpu().numpy() - g[f"s{i}_maps"]).max()))
        preds.append(lab.astype(np.int16)); gts.append(g[f"s{i}_gt"].astype(np.int16))
    hist = evaluate.hist_from_labels(gts, preds, 5)
    print(mode, "hist L1", int(np.abs(hist.cpu().numpy() - g["hist"]).sum()))
