import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle, oracle.vit, oracle.cam
from test_gpu_pipeline import TINY, tiny_model, dev, host, maxabs
from excel_amd import ops
g = np.load("/root/repo/tests/golden/lvc_tiny.npz")
model, w = tiny_model(g["text"].T.copy(), gemm_mode="f32")
wo = oracle.vit.reload_self_attn(w, TINY, 6, "train")
x = g["imgs"]; xc = np.concatenate([x, x[..., ::-1]], 0)
_, _, feats = oracle.vit.vit_forward(xc, wo, TINY)
out = model(dev(xc), want_feats=True)
gf = host(model.last_all_feats)
print("feats shape", gf.shape, feats.shape)
for l in range(gf.shape[0]): print(l, maxabs(gf[l], feats[l]) / np.abs(feats[l]).max())
ex_o = feats[-1][:, 1:, :24].transpose(0, 2, 1).reshape(4, 24, 6, 6)
ex_g = model.last_all_feats[-1][:, 1:, :24].permute(0, 2, 1).reshape(4, 24, 6, 6).contiguous()
print("ex feats", maxabs(host(ex_g), ex_o))
ea_g = host(ops.feature_affinity(ex_g, "mask_softmax")); ea_o = oracle.vit.ex_attention(ex_o)
print("ex_attn", maxabs(ea_g, ea_o), "flips", int(((ea_g == 0) != (ea_o == 0)).sum()))
ea_go = host(ops.feature_affinity(dev(ex_o), "mask_softmax"))
print("ex_attn same input", maxabs(ea_go, ea_o))
m_g = host(model(dev(xc), ex_feats=dev(ex_o)))
m_o = oracle.cam.attr_maps_raw(xc, wo, TINY, g["text"].T.copy(), 4, ex_feats=ex_o)[0]
print("maps same ex", maxabs(m_g, m_o))
