"""-m gpu: the train_hyper step against the CPU oracle (oracle/train.py, torch autograd)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import train as otrain                      # noqa: E402
from pcgcv1_amd import synthetic                          # noqa: E402
from pcgcv1_amd.train_hyper import Trainer                # noqa: E402


def _setup(seed=5, B=2, cs=16):
    w = synthetic.make_weights(seed=seed, profile="dense")
    # move the likelihoods away from their 1e-9 floor so that every gradient path is exercised
    w["hyper_decoder/conv4_2/bias"] = (w["hyper_decoder/conv4_2/bias"] + 0.8).astype(np.float32)
    x = synthetic.make_cubes(seed=seed, n_cubes=B, cube_size=cs, occupancy=0.06)
    rng = np.random.default_rng(seed)
    ny = (rng.random((B, cs // 4, cs // 4, cs // 4, 16)) - 0.5).astype(np.float32)
    nz = (rng.random((B, cs // 8, cs // 8, cs // 8, 8)) - 0.5).astype(np.float32)
    return w, x, ny, nz


def test_loss_terms_and_gradients_match_autograd():
    w, x, ny, nz = _setup()
    alpha, beta = 0.75, 3.0
    terms_ref, leaves = otrain.forward_loss(w, x, ny, nz, alpha, beta)
    tr = Trainer(w, alpha=alpha, beta=beta)
    terms = tr.forward_backward(x, ny, nz)
    for k in ("loss", "bpp_y", "bpp_z", "empty", "full"):
        assert abs(terms[k] - terms_ref[k]) <= 2e-4 * max(1.0, abs(terms_ref[k])), (k, terms[k], terms_ref[k])
    worst = []
    for name, leaf in leaves.items():
        g_ref = leaf.grad.numpy()
        g = tr.g[name].cpu().numpy()
        scale = float(np.abs(g_ref).max())
        assert scale > 0, name + ": oracle gradient is identically zero (test does not exercise it)"
        err = float(np.abs(g - g_ref).max()) / scale
        worst.append((err, name))
        assert err < 5e-3, (name, err, scale)
    print(sorted(worst)[-3:])


def test_adam_step_matches_tf1_form():
    w, x, ny, nz = _setup(seed=6)
    tr = Trainer(w, alpha=2.0, beta=3.0, lr=1e-3)
    tr.forward_backward(x, ny, nz)
    g = {k: v.cpu().numpy().copy() for k, v in tr.g.items()}
    tr.apply_gradients()
    tr.forward_backward(x, ny, nz)
    g2 = {k: v.cpu().numpy().copy() for k, v in tr.g.items()}
    tr.apply_gradients()
    new = tr.weights()
    for name in ("analysis_transform/conv_in/kernel", "hyper_decoder/conv4_2/bias", "estimator/matrix_1"):
        p, m, v = np.asarray(w[name], np.float32), 0.0, 0.0
        p, m, v = otrain.adam_step(p, g[name], m, v, 1, lr=1e-3)
        p, m, v = otrain.adam_step(p, g2[name], m, v, 2, lr=1e-3)
        np.testing.assert_allclose(new[name], p, rtol=2e-5, atol=1e-7)


def test_training_reduces_the_loss():
    w, x, ny, nz = _setup(seed=7)
    tr = Trainer(w, alpha=0.75, beta=3.0, lr=2e-4)
    first = tr.step(x, ny, nz)["loss"]
    for _ in range(5):
        last = tr.step(x, ny, nz)["loss"]
    assert np.isfinite(last) and last < first, (first, last)
