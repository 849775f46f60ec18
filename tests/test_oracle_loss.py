"""oracle/loss.py against hand-computed values and torch autograd (CPU): the reference has no test for loss.py."""
import numpy as np
import pytest

from oracle import loss as oloss

torch = pytest.importorskip("torch")


def test_bce_and_metrics_known_answers():
    pred = np.array([0.0, 2.0, -2.0, 30.0], np.float32).reshape(1, 2, 2, 1, 1)
    label = np.array([0.0, 1.0, 1.0, 0.0], np.float32).reshape(1, 2, 2, 1, 1)
    e, f = oloss.get_bce_loss(pred, label)
    sg = lambda v: 1 / (1 + np.exp(-v))
    one_minus = np.float32(1) - np.float32(1.0 - 1e-7)                            # sigmoid(30) clips at float32(1 - 1e-7) = 1 - 2^-23
    assert one_minus == np.float32(2.0 ** -23)
    assert abs(e - np.mean([-np.log(1 - sg(0.0)), -np.log(float(one_minus))])) < 2e-6
    assert abs(f - np.mean([-np.log(sg(2.0)), -np.log(sg(-2.0))])) < 1e-6
    tp, fp, fn = oloss.get_confusion_matrix(pred, label)                          # pred > 0: [F, T, F, T]
    assert tp.sum() == 1 and fp.sum() == 1 and fn.sum() == 1
    assert oloss.get_classify_metrics(pred, label) == (0.5, 0.5, 1 / 3)


def test_focal_loss_matches_torch_autograd():
    rng = np.random.default_rng(0)
    yp = rng.uniform(0.01, 0.99, 500)
    yt = (rng.random(500) > 0.8).astype(np.float64)
    p = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    t = torch.tensor(yt)
    pt1 = torch.clamp(torch.where(t == 1, p, torch.ones_like(p)), 1e-3, .999)
    pt0 = torch.clamp(torch.where(t == 0, p, torch.zeros_like(p)), 1e-3, .999)
    loss = -(0.9 * (1 - pt1) ** 2 * torch.log(pt1)).sum() - (0.1 * pt0 ** 2 * torch.log(1 - pt0)).sum()
    loss.backward()
    assert abs(oloss.get_focal_loss(yp, yt) - float(loss)) < 1e-4 * abs(float(loss))
    np.testing.assert_allclose(oloss.focal_loss_grad(yp, yt), p.grad.numpy(), rtol=1e-9, atol=1e-12)
