"""Bjontegaard deltas between two rate-distortion curves, as the reference's result notebooks use them
(myutils/bdrate_metrics.py:27-129: `bdsnr`, `bdrate` over lists of (rate, psnr) points).

Both are the mean gap between the two cubic least-squares fits over the interval the curves share:
  bdsnr   psnr as a cubic in ln(rate), averaged over the common ln(rate) range            -> dB
  bdrate  ln(rate) as a cubic in psnr, averaged over the common psnr range, as a rate change in per cent
          (the exponent is clipped at 200 like the reference, :121-122)
Pinned to the reference module's own outputs on seeded curves (tests/golden/bdrate.npz).
"""
import numpy as np


def _mean_gap(x1, y1, x2, y2):
    """Average over the common x-range of fit2(x) - fit1(x), each a degree-3 least-squares polynomial."""
    lo, hi = max(x1.min(), x2.min()), min(x1.max(), x2.max())
    if hi == lo:
        return None
    area = []
    for x, y in ((x1, y1), (x2, y2)):
        anti = np.polyint(np.polyfit(x, y, 3))
        area.append(np.polyval(anti, hi) - np.polyval(anti, lo))
    return float((area[1] - area[0]) / (hi - lo))


def _split(points):
    p = np.asarray(points, np.float64).reshape(-1, 2)
    return np.log(p[:, 0]), p[:, 1]


def bdsnr(metric_set1, metric_set2):
    (lr1, q1), (lr2, q2) = _split(metric_set1), _split(metric_set2)
    gap = _mean_gap(lr1, q1, lr2, q2)
    return 0.0 if gap is None else gap


def bdrate(metric_set1, metric_set2):
    (lr1, q1), (lr2, q2) = _split(metric_set1), _split(metric_set2)
    gap = _mean_gap(q1, lr1, q2, lr2)
    if gap is None:
        raise ZeroDivisionError("the two curves share a single psnr value")
    return (np.exp(min(gap, 200.0)) - 1.0) * 100.0
