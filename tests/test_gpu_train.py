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
