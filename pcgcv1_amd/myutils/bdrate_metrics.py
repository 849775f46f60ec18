"""Bjøntegaard deltas between two rate-distortion curves — same entry points as the reference's
myutils/bdrate_metrics.py (bdsnr 28-75, bdrate 78-129; used by results.ipynb for its BD tables):

  bdsnr(curve1, curve2)   average PSNR gain of curve 2 over curve 1 at equal rate, in dB
  bdrate(curve1, curve2)  average rate change of curve 2 against curve 1 at equal PSNR, in percent (negative = saves bits)

A curve is a sequence of (rate, psnr) pairs.  Both measures fit a cubic through one coordinate as a function of the
other — PSNR over ln(rate) for bdsnr, ln(rate) over PSNR for bdrate — and average the difference of the two cubics over
the interval the curves share.  Host-side reporting (a handful of points): numpy only.
"""
import numpy as np


def _mean_gap(x1, y1, x2, y2):
    """Mean over the common x interval of cubic2(x) - cubic1(x), each cubic a least-squares fit through (x_i, y_i).
    A common interval of length zero gives None."""
    lo, hi = max(np.min(x1), np.min(x2)), min(np.max(x1), np.max(x2))
    if hi == lo:
        return None
    area = []
    for x, y in ((x1, y1), (x2, y2)):
        primitive = np.polyint(np.polyfit(x, y, 3))
        area.append(np.polyval(primitive, hi) - np.polyval(primitive, lo))
    return (area[1] - area[0]) / (hi - lo)


def _split(curve):
    pts = np.asarray(list(curve), np.float64).reshape(-1, 2)
    return np.log(pts[:, 0]), pts[:, 1]


def bdsnr(metric_set1, metric_set2):
    """bdrate_metrics.py:28-75.  Curves that share a single rate give 0.0, as the reference does."""
    r1, p1 = _split(metric_set1)
    r2, p2 = _split(metric_set2)
    gap = _mean_gap(r1, p1, r2, p2)
    return 0.0 if gap is None else float(gap)


def bdrate(metric_set1, metric_set2):
    """bdrate_metrics.py:78-129: the mean gap of ln(rate) is capped at 200 before it becomes a percentage; curves that
    share a single PSNR value are 0 / 0 in the reference (nan, with numpy's warning) and nan here."""
    r1, p1 = _split(metric_set1)
    r2, p2 = _split(metric_set2)
    gap = _mean_gap(p1, r1, p2, r2)
    if gap is None:
        return float("nan")
    return float((np.exp(min(gap, 200.0)) - 1.0) * 100.0)
