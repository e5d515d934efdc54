"""Dev soak: random shapes / class sets through the batched pipeline vs the oracle (exact-fp32 mode, tiny ViT)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from oracle.vit import VitConfig, make_vit_weights
from excel_amd.model import ExCEL_model
from excel_amd.pipeline import TrainingFreePipeline
TINY = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)
KW = dict(width=128, layers=8, heads=2, patch=16, output_dim=64, input_resolution=64)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    rs = np.random.RandomState(1000 + case)
    S = int(rs.choice([64, 96, 128, 160]))
    B = int(rs.randint(1, 5))
    F_ = int(rs.randint(2, 7)); T = F_ + int(rs.randint(1, 6))
    H, W = int(rs.randint(20, 150)), int(rs.randint(20, 150))
    w = make_vit_weights(TINY, seed=int(rs.randint(0, 100)))
    text = rs.standard_normal((T, 64)).astype(np.float32); text /= np.linalg.norm(text, axis=1, keepdims=True)
    model = ExCEL_model(clip_model="tiny", num_classes=F_ + 1, img_size=S, mode="train", state_dict=w, vit_cfg=KW, text_attr=text.T.copy(), gemm_mode=os.environ.get("SOAK_MODE", "f32"))
    wo = oracle.vit.reload_self_attn(w, TINY, S // 16, "train")
    imgs = rs.standard_normal((B, 3, S, S)).astype(np.float32)
    gts = rs.randint(0, F_ + 1, (B, H, W)).astype(np.uint8); gts[rs.rand(B, H, W) < 0.03] = 255
    cls = np.zeros((B, F_), np.float32)
    for b in range(B):
        cls[b, rs.choice(F_, size=int(rs.randint(1, min(F_, 4) + 1)), replace=False)] = 1
    smax = int(cls.sum(1).max())
    thr = float(rs.choice([0.79, 0.88, 0.5]))
    pipe = TrainingFreePipeline(model, num_classes=F_ + 1, smax=smax, caa_thre=thr)
    labels = pipe.run_batch(torch.from_numpy(imgs).cuda(), torch.from_numpy(cls).cuda(), torch.from_numpy(gts).cuda())
    lab = labels.cpu().numpy()
    par = oracle.par.PAR([1, 2, 4, 8, 12, 24], 20)
    agree = []
    ref_hist = np.zeros((F_ + 1, F_ + 1), np.int64)
    for b in range(B):
        r = oracle.pipeline.run_sample(imgs[b], cls[b], (H, W), wo, TINY, text.T.copy(), F_, par, S, caa_thre=thr)
        agree.append(float(np.mean(lab[b] == r)))
        ref_hist += oracle.evaluate.fast_hist(gts[b].flatten(), lab[b].flatten(), F_ + 1)
    ok = min(agree) >= 0.995 and np.array_equal(pipe.hist.cpu().numpy(), ref_hist)
    bad += 0 if ok else 1
    print(f"case {case:2d} S={S} B={B} F={F_} T={T} HxW={H}x{W} thr={thr}: min agreement {min(agree):.4f} hist_ok={np.array_equal(pipe.hist.cpu().numpy(), ref_hist)}", "" if ok else "  <<<<<")
print("bad cases:", bad)
