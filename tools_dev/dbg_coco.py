import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle, oracle.vit, oracle.aff, oracle.cam, oracle.pipeline, oracle.par
from test_gpu_pipeline import SMALL512, SMALL512_KW, dev, host, maxabs, make_vit_weights
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
from excel_amd import ops
import os
mode = os.environ.get("MODE", "bf16x3")
rs = np.random.RandomState(5)
w = make_vit_weights(SMALL512, seed=3)
text = rs.standard_normal((103, 64)).astype(np.float32); text /= np.linalg.norm(text, axis=1, keepdims=True)
model = ExCEL_model(clip_model="small", num_classes=81, img_size=512, mode="train", state_dict=w, vit_cfg=SMALL512_KW, text_attr=text.T.copy(), gemm_mode=mode)
wo = oracle.vit.reload_self_attn(w, SMALL512, 32, "train")
B, S, F = 2, 512, 80
imgs = rs.standard_normal((B, 3, S, S)).astype(np.float32)
gts = rs.randint(0, 81, (B, S, S)).astype(np.uint8)
cls = np.zeros((B, F), np.float32); cls[0, [3, 17, 40, 41, 79]] = 1; cls[1, [0, 62]] = 1
pipe = TrainingFreePipeline(model, num_classes=81, smax=5, caa_thre=0.88)
labels, inter = pipe.run_batch(dev(imgs), dev(cls), dev(gts), return_intermediates=True)
par = oracle.par.PAR([1, 2, 4, 8, 12, 24], 20)
for b in range(B):
    r = oracle.pipeline.run_sample(imgs[b], cls[b], (S, S), wo, SMALL512, text.T.copy(), F, par, S, caa_thre=0.88, return_all=True)
    k = int(cls[b].sum())
    attn = r["attn_weights"]
    print("attn shape", attn.shape)
    waff_ref = attn[-8:, 0].mean(0)[1:, 1:]
    wa = host(inter["w_aff"])[b]
    print("w_aff", wa.shape, maxabs(wa, waff_ref) if waff_ref is not None else None, np.abs(waff_ref).max())
    ref = np.asarray(r["refined"]); got = host(inter["refined"])[b, :k]
    print("refined shapes", ref.shape, got.shape)
    for c in range(k):
        print(" class", c, "maxabs", maxabs(got[c].reshape(-1), ref[c].reshape(-1)), "ref max", ref[c].max())
    # masks
    cl = r["cls_lst"]
    maps = r["attr_maps_raw"][0]
    for c, ci in enumerate(cl):
        cam = maps[:, ci].reshape(32, 32)
        bx = oracle.aff.scoremap2bbox(cam, 0.88, multi_contour_eval=True) if hasattr(oracle.aff, "scoremap2bbox") else None
        print("  boxes", c, bx)
