"""Pin the numpy oracle against goldens captured from the reference's own Python
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

import oracle
from oracle.vit import VitConfig, make_vit_weights

TINY = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def relmax(a, b):
    """max-abs error relative to the reference's largest magnitude (fp32 round-off is
    amplified ~100x through the 8 gain-4 attention blocks of the tiny model)."""
    return maxabs(a, b) / float(np.max(np.abs(b)))


@pytest.mark.parametrize("mode", ["train", "val"])
def test_vit_forward_and_cam(golden, mode):
    g = golden("vit_cam_tiny.npz")
    w = make_vit_weights(TINY, seed=int(g["seed_w"]))
    w = oracle.vit.reload_self_attn(w, TINY, feat_size=6, mode=mode)
    f, attn, feats = oracle.cam.generate_clip_fts(g["imgs"], w, TINY)
    x, _, _ = oracle.vit.vit_forward(g["imgs"], w, TINY)
    assert relmax(x, g[f"{mode}_x"]) < 2e-5
    assert maxabs(f, g[f"{mode}_image_features"]) < 1e-5
    assert attn.shape == g[f"{mode}_attn"].shape
    assert maxabs(attn, g[f"{mode}_attn"]) < 1e-4      # values up to heads=2; fp32 through 8 blocks
    # block-6-style (index 2) weights are head-averaged, surgery ones head-summed (quirk Q3)
    np.testing.assert_allclose(attn[2].sum(-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(attn[5].sum(-1), TINY.heads, atol=1e-5)
    assert relmax(feats[0], g[f"{mode}_feat0"]) < 2e-5
    assert relmax(feats[-1], g[f"{mode}_feat_last"]) < 5e-5
    cam = oracle.cam.clip_feature_surgery(f, g[f"{mode}_text"])
    assert maxabs(cam, g[f"{mode}_cam"]) < 2e-5


def test_vit_second_resolution_uses_resized_grid(golden):
    g = golden("vit_cam_tiny.npz")
    w = make_vit_weights(TINY, seed=int(g["seed_w"]))
    w = oracle.vit.reload_self_attn(w, TINY, feat_size=6, mode="train")
    x, attn, _ = oracle.vit.vit_forward(g["res2_imgs"], w, TINY)
    assert relmax(x, g["res2_x"]) < 2e-5
    assert maxabs(attn[-1], g["res2_attn_last"]) < 1e-4


def test_clip_feature_surgery(golden):
    g = golden("ops.npz")
    out = oracle.cam.clip_feature_surgery(g["cfs_f"], g["cfs_t"])
    assert maxabs(out, g["cfs_out"]) < 1e-5


def test_compute_trans_mat(golden):
    g = golden("ops.npz")
    out = oracle.aff.compute_trans_mat(g["tm_in"])
    assert maxabs(out, g["tm_out"]) < 1e-7
    np.testing.assert_allclose(out, g["tm_out"], rtol=2e-5)


def test_par(golden):
    g = golden("ops.npz")
    par = oracle.par.PAR([1, 2, 4, 8, 12, 24], 20)
    out = par(g["par_img"], g["par_mask"])
    assert out.shape == g["par_out"].shape
    assert maxabs(out, g["par_out"]) < 2e-5
    par3 = oracle.par.PAR([1, 2, 4, 8, 12, 24], 3)
    out2 = par3(g["par2_img"], g["par2_mask"])
    assert maxabs(out2, g["par2_out"]) < 1e-5


def test_par_position_softmax_constants():
    # std48(pos) = 9.9003 ; d=1 straight tap weight 0.07645
    par = oracle.par.PAR([1, 2, 4, 8, 12, 24], 1)
    assert abs(float(par.pos.std(ddof=1)) - 9.9003) < 1e-3
    z = -((par.pos / (par.pos.std(ddof=1) + 1e-8) / 0.3) ** 2)
    sm = np.exp(z - z.max())
    sm /= sm.sum()
    assert abs(sm[1] - 0.07645) < 1e-4 and abs(sm[0] - 0.06826) < 1e-4


def test_scores(golden):
    g = golden("ops.npz")
    hist = oracle.evaluate.hist_of(list(g["sc_gts"]), list(g["sc_preds"]), 21)
    assert np.array_equal(hist, g["sc_hist"])
    sc = oracle.evaluate.scores(list(g["sc_gts"]), list(g["sc_preds"]), 21)
    assert abs(sc["miou"] - float(g["sc_miou"])) < 1e-12
    assert abs(sc["pAcc"] - float(g["sc_pacc"])) < 1e-12
    assert abs(sc["mAcc"] - float(g["sc_macc"])) < 1e-12
    np.testing.assert_allclose(np.array(list(sc["iou"].values())), g["sc_iou"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(list(sc["precision"].values())), g["sc_prec"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(list(sc["recall"].values())), g["sc_rec"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(list(sc["confusion"].values())), g["sc_conf"], rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("ds,F,K", [("pascal_voc", 20, 112), ("ms_coco", 80, 224)])
def test_attr_aggregate(golden, ds, F, K):
    g = golden("attr_aggregate.npz")
    bank = golden(f"attr_bank_{ds}.npz")["bank"]
    assert bank.shape == (512, K)
    out = oracle.attr.attr_aggregate(g[f"{ds}_text"], bank, F)
    assert out.shape == g[f"{ds}_agg"].shape
    assert maxabs(out, g[f"{ds}_agg"]) < 1e-6
    np.testing.assert_allclose(np.linalg.norm(out, axis=0), 1.0, atol=1e-5)


def test_pipeline_trace(golden):
    g = golden("pipeline_tiny.npz")
    w = make_vit_weights(TINY, seed=int(g["seed_w"]))
    w = oracle.vit.reload_self_attn(w, TINY, feat_size=6, mode="train")
    text_attr = g["text"].T.copy()
    par = oracle.par.PAR([1, 2, 4, 8, 12, 24], 20)
    samples = []
    for i in range(4):
        r = oracle.pipeline.run_sample(g[f"s{i}_img"], g[f"s{i}_cls"], g[f"s{i}_gt"].shape, w, TINY,
                                       text_attr, 4, par, 96, return_all=True)
        assert maxabs(r["inputs"], g[f"s{i}_inputs"]) < 1e-5
        assert maxabs(r["attr_maps_raw"], g[f"s{i}_maps"]) < 5e-5
        assert np.array_equal(r["cls_lst"], g[f"s{i}_cls_lst"])
        assert maxabs(np.stack(r["refined"]), g[f"s{i}_refined"]) < 1e-5
        assert maxabs(r["cams"], g[f"s{i}_cams"]) < 5e-5
        agree = np.mean(r["label"] == g[f"s{i}_label"])
        assert agree >= 0.999, agree
        samples.append((g[f"s{i}_img"], g[f"s{i}_gt"], g[f"s{i}_cls"]))
    hist, _ = oracle.pipeline.build_validation(samples, w, TINY, text_attr, num_classes=5, resize_size=96)
    assert np.abs(hist - g["hist"]).sum() <= 4          # argmax near-ties only
    assert abs(oracle.evaluate.scores_from_hist(hist)["miou"] - float(g["miou"])) < 1e-3


def test_lvc_branch_ex_feats_and_seg_attn(golden):
    """ex_feats branch of the surgery attention (clip_surgery_model.py:127-141), attn_pred
    (model_excel.py:70-76) and the seg_attn layer selection (affutils.py:182-195)."""
    g = golden("lvc_tiny.npz")
    w = make_vit_weights(TINY, seed=int(g["seed_w"]))
    w = oracle.vit.reload_self_attn(w, TINY, feat_size=6, mode="train")
    x, attn, _ = oracle.vit.vit_forward(g["imgs"], w, TINY, ex_feats=g["ex_feats"])
    assert relmax(x, g["x"]) < 2e-5
    assert maxabs(attn, g["attn"]) < 1e-4
    f, _, _ = oracle.cam.generate_clip_fts(g["imgs"], w, TINY, ex_feats=g["ex_feats"])
    cam = oracle.cam.clip_feature_surgery(f, g["text"])
    assert maxabs(cam, g["cam"]) < 2e-5
    # the branch really changes the result (guards against a silently ignored argument)
    x0, _, _ = oracle.vit.vit_forward(g["imgs"], w, TINY)
    assert relmax(x0, g["x"]) > 1e-3
    ap = oracle.cam.attn_pred(g["ex_feats"])
    assert maxabs(ap, g["attn_pred"]) < 1e-6
    refined, cls_lst = oracle.aff.refine_cams_with_aff(g["cam"][0, 1:, :4], g["attn"][:, 0], g["cls"], size=(96, 96),
                                                       caa_thre=0.79, seg_attn=g["attn_pred"][0][None])
    assert list(cls_lst) == list(g["cls_lst"])
    assert maxabs(np.stack(refined, 0), g["refined"]) < 1e-6


def test_decoder_head_and_aliased_feats(golden):
    """what the decoder receives (all_feats with the reference's in-place aliasing, quirk Q4), the SegFormer
    fuse, the 3-layer decoder transformer and attn_pred, against the reference's own modules."""
    g = golden("decoder_tiny.npz")
    w = make_vit_weights(TINY, seed=int(g["seed_w"]))
    w = oracle.vit.reload_self_attn(w, TINY, feat_size=6, mode="train")
    _, _, feats = oracle.vit.vit_forward(g["imgs"], w, TINY, aliased_feats=True)
    for l in range(TINY.layers):
        assert relmax(feats[l], g["all_feats"][l]) < 5e-5, l
    _, _, clean = oracle.vit.vit_forward(g["imgs"], w, TINY)
    assert relmax(clean[2], g["all_feats"][2]) > 1e-2          # the quirk is real: entry 2 (last single-path block) differs
    dw = {k: g[k] for k in g.files if k.startswith(("fuse.", "dec."))}
    fts = oracle.decoder.segformer_fuse(g["all_feats"], dw)
    assert relmax(fts, g["fts"]) < 1e-5
    seg, attns = oracle.decoder.decoder_transformer(g["fts"], dw, heads=8)
    assert relmax(seg, g["seg"]) < 1e-5
    assert maxabs(attns[-1], g["dec_attn_last"]) < 1e-6
    assert maxabs(oracle.cam.attn_pred(g["fts"]), g["attn_pred"]) < 1e-6


def test_text_tower_and_prompt_ensemble(golden):
    """encode_text (causal transformer, EOT row, projection) and the prompt-ensemble reduction."""
    g = golden("text_tiny.npz")
    w = {k[2:]: g[k] for k in g.files if k.startswith("w.")}
    out = oracle.text.encode_text(g["tokens"], w, heads=2)
    assert relmax(out, g["out"]) < 1e-5
    assert maxabs(oracle.text.prompt_ensemble(g["out"]), g["ensemble"]) < 1e-6


def test_torch_cpu_port_matches_numpy_oracle():
    """oracle/torch_cpu.py (the multi-threaded CPU baseline of bench.py) against the numpy oracle it mirrors: ViT features / last-6
    attention weights to fp32 round-off, PAR, and the batch-1 evaluation loop label for label."""
    from oracle import torch_cpu
    from oracle.vit import VitConfig, make_vit_weights
    cfg = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)
    w = oracle.vit.reload_self_attn(make_vit_weights(cfg, seed=11), cfg, 6, "train")
    rs = np.random.RandomState(3)
    text = rs.standard_normal((9, 64)).astype(np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    samples = []
    for i in range(3):
        cls = np.zeros(4, np.float32)
        cls[rs.choice(4, size=1 + i % 2, replace=False)] = 1
        samples.append((rs.standard_normal((3, 96, 96)).astype(np.float32), rs.randint(0, 5, (70, 90)).astype(np.uint8), cls))
    x1, a1, _ = oracle.vit.vit_forward_single(samples[0][0], w, cfg)
    x2, a2 = torch_cpu.TorchVit(w, cfg, 6).forward(samples[0][0])
    assert np.abs(x1 - x2).max() < 1e-4 and np.abs(np.stack(a1[-6:]) - a2).max() < 1e-4
    m = rs.rand(1, 3, 31, 45).astype(np.float32)
    im = rs.standard_normal((1, 3, 24, 24)).astype(np.float32)
    assert np.abs(oracle.par.PAR([1, 2, 4, 8, 12, 24], 5)(im, m) - torch_cpu.TorchPAR([1, 2, 4, 8, 12, 24], 5)(im, m)).max() < 1e-5
    h1, p1 = oracle.pipeline.build_validation(samples, w, cfg, text.T.copy(), num_classes=5, resize_size=96)
    h2, p2, stage = torch_cpu.build_validation(iter(samples), w, cfg, text.T.copy(), num_classes=5, resize_size=96)
    assert np.array_equal(h1, h2) and all(np.array_equal(a, b) for a, b in zip(p1, p2))
    assert set(stage) == {"vit", "cam", "aff_random_walk", "upsample_par_argmax", "score"}


def test_dcrf_oracle_properties():
    """oracle/dcrf.py restates the published permutohedral-lattice mean field (no pydensecrf here: parity unpinned): barycentric weights
    are a partition of unity, the lattice filter is linear and nearly self-adjoint, marginals stay normalised, a strong Potts
    term smooths an isolated outlier, and zero iterations is softmax(-U)."""
    from oracle import dcrf
    rs = np.random.RandomState(0)
    H, W, C = 20, 26, 3
    img = (rs.rand(H, W, 3) * 40 + 100).astype(np.uint8)
    lat = dcrf.Permutohedral(dcrf.features_2d(H, W, 67, img, 3))
    np.testing.assert_allclose(lat.bary.sum(1), 1.0, atol=2e-5)
    assert lat.bary.min() > -1e-5 and lat.offset.max() == lat.M - 1
    a, b = rs.rand(H * W, 2).astype(np.float32), rs.rand(H * W, 2).astype(np.float32)
    # (near-)symmetry: the transpose of the filter applies the blur passes in reverse order, which differs only where a neighbour is absent
    assert abs(float((a * lat.compute(b)).sum()) - float((lat.compute(a) * b).sum())) < 1e-2 * float((a * lat.compute(b)).sum())
    np.testing.assert_allclose(lat.compute(2 * a + b), 2 * lat.compute(a) + lat.compute(b), rtol=1e-4, atol=1e-5)
    p = rs.rand(C, H, W).astype(np.float32)
    p /= p.sum(0, keepdims=True)
    crf = dcrf.DenseCRF(iter_max=10, pos_w=3, pos_xy_std=1, bi_w=4, bi_xy_std=67, bi_rgb_std=3)     # tools/infer_lam.py:191-198
    Q = crf(img, p)
    np.testing.assert_allclose(Q.sum(0), 1.0, atol=1e-5)
    Q0 = dcrf.dense_crf_2d(img, dcrf.unary_from_softmax(p), 0, 3, 1, 4, 67, 3)
    np.testing.assert_allclose(Q0, p / p.sum(0, keepdims=True), atol=1e-5)        # softmax(log p) = p
    # one confident outlier inside a uniform, uniformly coloured region is absorbed by its neighbours
    flat = np.full((H, W, 3), 128, np.uint8)
    pp = np.zeros((2, H, W), np.float32)
    pp[0] = 0.7
    pp[1] = 0.3
    pp[:, 10, 13] = (0.3, 0.7)
    assert dcrf.DenseCRF(10, 3, 1, 4, 67, 3)(flat, pp).argmax(0)[10, 13] == 0
    lab = dcrf.crf_inference_label(flat, rs.randint(0, 3, (H, W)), t=2, n_labels=3)
    assert lab.shape == (H, W)
