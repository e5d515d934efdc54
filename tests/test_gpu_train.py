"""GPU parity tests of the training iteration against goldens minted from the reference's own modules, losses,
affinity-label code and optimizer through autograd (tests/golden/make_goldens.py gold_train)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) / max(float(np.max(np.abs(b))), 1e-30)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from excel_amd import ops as _ops
    return _ops


def test_losses_and_their_gradients(ops, golden):
    g = golden("train_tiny.npz")
    losses, d_seg, d_ap = ops.train_losses(dev(g["seg"]), dev(g["attn_pred"]), dev(g["pseudo"]), radius=2, w_seg=1.0, w_diver=0.1)
    l = host(losses)
    assert abs(l[0] - float(g["seg_loss"])) < 2e-6 * max(1.0, float(g["seg_loss"]))
    assert abs(l[1] - float(g["diver_loss"])) < 2e-6
    assert relmax(host(d_seg), g["d_seg"]) < 2e-5
    assert relmax(host(d_ap), g["d_attn_pred"]) < 1e-6
    # affinity label structure: zero gradient exactly where the reference's mask says "ignore"
    assert np.array_equal(host(d_ap) == 0, g["aff_mask"] == 255)


def _handle_and_names(ops, g, tag="w0"):
    fuse_sd = {k[len(tag) + 6:]: g[k] for k in g.files if k.startswith(tag + ".fuse.")}
    dec_sd = {k[len(tag) + 5:]: g[k] for k in g.files if k.startswith(tag + ".dec.")}
    h = ops.DecoderHandle(fuse_sd, dec_sd, heads=8)
    names = {}                                            # handle key -> golden suffix
    for l in range(h.cfg["vit_layers"]):
        for f, k in (("proj_w", "proj.weight"), ("proj_b", "proj.bias"), ("proj2_w", "proj_2.weight"), ("proj2_b", "proj_2.bias")):
            names[f"fuse{l}.{f}"] = f"fuse.linears_modulelist.{l}.{k}"
    for l in range(h.cfg["dec_layers"]):
        for f, k in ops._BLOCK_KEYS.items():
            names[f"blk{l}.{f}"] = f"dec.transformer.resblocks.{l}.{k}"
    names.update({"fuse_w": "fuse.linear_fuse.weight", "fuse_b": "fuse.linear_fuse.bias", "pred_w": "dec.linear_pred.weight",
                  "pred_b": "dec.linear_pred.bias"})
    return h, names


def test_decoder_backward_matches_autograd(ops, golden):
    """Gradient of every decoder parameter for one iteration (train_voc.py:186-218) vs the reference modules under autograd."""
    g, gd = golden("train_tiny.npz"), golden("decoder_tiny.npz")
    h, names = _handle_and_names(ops, g)
    seg, ap, ctx = h.forward_train(dev(gd["all_feats"]))
    assert relmax(host(seg), g["seg"]) < 2e-5 and relmax(host(ap), g["attn_pred"]) < 1e-5
    losses, d_seg, d_ap = ops.train_losses(seg, ap, dev(g["pseudo"]), radius=2, w_seg=1.0, w_diver=0.1)
    assert abs(float(losses[0]) - float(g["seg_loss"])) < 1e-5 and abs(float(losses[1]) - float(g["diver_loss"])) < 1e-5
    grads = h.backward(ctx, d_seg, d_ap)
    worst = ("", 0.0)
    for k, suffix in names.items():
        ref = g["g." + suffix].reshape(host(grads[k]).shape)
        e = relmax(host(grads[k]), ref)
        if e > worst[1]:
            worst = (k, e)
        assert e < 2e-4, (k, e)
    print("worst gradient mismatch", worst)


def test_two_adamw_steps_match_reference_optimizer(ops, golden):
    """Two full iterations (forward, losses, backward, PolyWarmupAdamW with the reference's schedule) -> the reference's parameters."""
    g, gd = golden("train_tiny.npz"), golden("decoder_tiny.npz")
    h, names = _handle_and_names(ops, g)
    lr0, warm, ratio = 1e-3, 50, 1e-6                                   # group 3: lr = args.lr * 10 (engine/optimizer_engine.py)
    for it in range(2):
        seg, ap, ctx = h.forward_train(dev(gd["all_feats"]))
        losses, d_seg, d_ap = ops.train_losses(seg, ap, dev(g["pseudo"]), radius=2, w_seg=1.0, w_diver=0.1)
        h.backward(ctx, d_seg, d_ap)
        lr = lr0 * (1 - (1 - it / warm) * (1 - ratio))                   # utils/optimizer.py PolyWarmupAdamW.step, warm-up branch
        assert abs(lr - float(g[f"lr_it{it}"])) < 1e-12
        h.adamw_step(lr, it + 1, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
        for k, suffix in names.items():
            ref = g[f"w{it + 1}." + suffix].reshape(host(h.t[k]).shape)
            assert float(np.max(np.abs(host(h.t[k]) - ref))) < 2e-6 * max(1.0, float(np.abs(ref).max())), (it, k)
    assert abs(float(losses[0]) - float(g["seg_loss_it1"])) < 1e-5
    # checkpoint round trip in the reference's key/shape convention
    fuse_sd, dec_sd = h.state_dicts()
    assert tuple(fuse_sd["linear_fuse.weight"].shape) == g["w2.fuse.linear_fuse.weight"].shape
    assert tuple(dec_sd["linear_pred.weight"].shape) == g["w2.dec.linear_pred.weight"].shape
    h2 = ops.DecoderHandle({k: host(v) for k, v in fuse_sd.items()}, {k: host(v) for k, v in dec_sd.items()}, heads=8)
    s1, _, _ = h.forward_train(dev(gd["all_feats"]))
    s2, _, _ = h2.forward_train(dev(gd["all_feats"]))
    assert torch.equal(s1, s2)


def test_train_step_end_to_end(ops, golden):
    """DecoderTrainer.train_step (scripts/train_voc.py:172-220): frozen tower + CAMs + pseudo labels + losses + backward + AdamW,
    in both regimes (before / after the LVC switch); the loss goes down on a repeated batch and every step is reproducible."""
    from oracle.vit import VitConfig, make_vit_weights
    from excel_amd.model import ExCEL_model
    from excel_amd.scripts.train_voc import DecoderTrainer, poly_warmup_lr
    from excel_amd.utils.PAR import PAR
    g = golden("train_tiny.npz")
    TINY = VitConfig(width=128, layers=8, heads=2, patch=16, out_dim=64, input_resolution=64, n_surgery=5)
    kw = dict(width=128, layers=8, heads=2, patch=16, output_dim=64, input_resolution=64)
    w = make_vit_weights(TINY, seed=11)
    rs = np.random.RandomState(3)
    text = rs.standard_normal((9, 64)).astype(np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    sd = {"decoder_fts_fuse." + k[8:]: g[k] for k in g.files if k.startswith("w0.fuse.")}
    sd.update({"decoder." + k[7:]: g[k] for k in g.files if k.startswith("w0.dec.")})

    def run(lvc_iter, dropout_p=0.0, seg_aff_iter=10 ** 9):
        model = ExCEL_model(clip_model="tiny", num_classes=5, img_size=96, mode="train", state_dict=w, vit_cfg=kw, text_attr=text.T.copy(),
                            gemm_mode="f32", embedding_dim=32, in_channels=128, decoder_state_dict=sd)
        tr = DecoderTrainer(model, PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]), lr=1e-3, warmup_iters=2, max_iters=100, radius=2, lvc_iter=lvc_iter, dropout_p=dropout_p, seg_aff_iter=seg_aff_iter)
        x = dev(rs.standard_normal((2, 3, 96, 96)).astype(np.float32) * 0 + np.random.RandomState(5).standard_normal((2, 3, 96, 96)).astype(np.float32))
        cls = dev(np.array([[1, 0, 1, 0], [0, 1, 0, 0]], np.float32))
        return [tr.train_step(x, cls) for _ in range(6)], model
    hist, _ = run(lvc_iter=10 ** 9)
    assert all(np.isfinite(h["seg_loss"]) and np.isfinite(h["diver_loss"]) for h in hist)
    assert hist[-1]["seg_loss"] < hist[0]["seg_loss"]                   # same batch six times: the head fits its pseudo labels
    assert abs(hist[3]["lr"] - poly_warmup_lr(1e-2, 3, 2, 100, 1e-6, 1)) < 1e-12
    hist2, _ = run(lvc_iter=10 ** 9)
    assert [h["seg_loss"] for h in hist] == [h["seg_loss"] for h in hist2]        # fixed-order reductions: bit-reproducible
    hist3, _ = run(lvc_iter=2)                                                     # LVC regime from iteration 2 on (ex_feats CAMs + seg_attn)
    assert all(np.isfinite(h["seg_loss"]) for h in hist3) and hist3[0]["seg_loss"] == hist[0]["seg_loss"]
    assert tuple(hist3[-1]["aff_pseudos"].shape) == (2, 96, 96)
    hist4, _ = run(lvc_iter=10 ** 9, dropout_p=0.1, seg_aff_iter=3)                # Dropout2d on, affinity target = seg arg-max from iteration 3
    hist5, _ = run(lvc_iter=10 ** 9, dropout_p=0.1, seg_aff_iter=3)
    assert [h["seg_loss"] for h in hist4] == [h["seg_loss"] for h in hist5]        # the counter-based mask keeps it reproducible
    assert hist4[0]["seg_loss"] != hist[0]["seg_loss"] and all(np.isfinite(h["diver_loss"]) for h in hist4)


def test_train_attn_fts_is_the_post_dropout_cue(ops, golden):
    """scripts/train_voc.py:186-189: fts_diver = attn_fts.clone().detach() of the TRAIN-mode forward, i.e. after the head's Dropout2d.
    Without dropout it equals the inference head's attn_fts; with p > 0 whole (image, channel) planes are zero and the others are the
    inference values scaled by 1 / (1 - p)."""
    g, gd = golden("train_tiny.npz"), golden("decoder_tiny.npz")
    h, _ = _handle_and_names(ops, g)
    feats = dev(gd["all_feats"])
    ref = host(h.forward(feats, want_seg=False)[0])                     # [B,E,g,g]
    _, _, ctx0 = h.forward_train(feats)
    np.testing.assert_allclose(host(h.train_attn_fts(ctx0)), ref, rtol=0, atol=1e-6 * float(np.abs(ref).max()))
    p = 0.25
    _, _, ctx = h.forward_train(feats, dropout_p=p, dropout_seed=7)
    got = host(h.train_attn_fts(ctx))
    dead = np.all(got == 0, axis=(2, 3))                                # [B,E] dropped planes
    assert 0 < dead.sum() < dead.size and abs(dead.mean() - p) < 0.2
    keep = ~dead
    np.testing.assert_allclose(got[keep], ref[keep] / (1 - p), rtol=0, atol=2e-6 * float(np.abs(ref).max()) / (1 - p))
    with pytest.raises(ValueError):
        h.forward_train(feats[:, :, :, :-4].contiguous())              # wrong width: refused instead of reading out of bounds
