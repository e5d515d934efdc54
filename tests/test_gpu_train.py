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
