"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

numpy restatement of the reference's loss.py (TensorFlow ops replaced one for one):
  loss.py:8-33   get_bce_loss            loss.py:35-58  get_confusion_matrix
  loss.py:60-78  get_classify_metrics    loss.py:83-93  get_focal_loss
No reference test pins these (loss.py:127-167 only prints): *** PARITY UNPINNED ***."""
import numpy as np

F32 = np.float32


def get_bce_loss(pred, label):
    pred, label = np.asarray(pred, F32), np.asarray(label, F32)
    occupancy = np.clip(F32(1) / (F32(1) + np.exp(-pred)), F32(1e-7), F32(1.0 - 1e-7)).astype(F32)     # :18
    lab = label.max(axis=-1)
    occ = occupancy[..., 0]
    neg, pos = occ[lab == 0], occ[lab > 0]                                                               # :21-28
    with np.errstate(invalid="ignore"):
        empty = np.mean(-np.log(F32(1.0) - neg), dtype=np.float64) if neg.size else float("nan")
        full = np.mean(-np.log(pos), dtype=np.float64) if pos.size else float("nan")
    return float(empty), float(full)


def get_confusion_matrix(pred, label, th=0.):
    pred, label = np.asarray(pred, F32)[..., 0], np.asarray(label, F32)[..., 0]                         # :48-49
    p, l = (pred > th).astype(F32), (label > th).astype(F32)
    return p * l, p * (1 - l), (1 - p) * l                                                               # :54-56


def get_classify_metrics(pred, label, th=0.):
    tp, fp, fn = (float(m.sum(dtype=np.float64)) for m in get_confusion_matrix(pred, label, th))
    with np.errstate(invalid="ignore", divide="ignore"):
        return (float(np.float64(tp) / (tp + fp)), float(np.float64(tp) / (tp + fn)), float(np.float64(tp) / (tp + fp + fn)))


def get_focal_loss(y_pred, y_true, gamma=2, alpha=0.9):
    y_pred, y_true = np.asarray(y_pred, F32), np.asarray(y_true, F32)
    pt_1 = np.clip(np.where(y_true == 1, y_pred, F32(1)), F32(1e-3), F32(.999)).astype(F32)             # :87, :90
    pt_0 = np.clip(np.where(y_true == 0, y_pred, F32(0)), F32(1e-3), F32(.999)).astype(F32)             # :88, :91
    a = (F32(alpha) * np.power(F32(1.) - pt_1, F32(gamma)) * np.log(pt_1)).astype(F32)
    b = (F32(1 - alpha) * np.power(pt_0, F32(gamma)) * np.log(F32(1.) - pt_0)).astype(F32)
    return float(-a.sum(dtype=np.float64) - b.sum(dtype=np.float64))                                     # :93


def focal_loss_grad(y_pred, y_true, gamma=2, alpha=0.9):
    """Analytic d loss / d y_pred in float64 (tf.clip_by_value passes no gradient outside its range)."""
    p, t = np.asarray(y_pred, np.float64), np.asarray(y_true, np.float64)
    inside = (p >= 1e-3) & (p <= .999)
    p = np.clip(p, 1e-3, .999)                      # only to keep the masked-out branches finite
    g1 = -alpha * ((1 - p) ** gamma / p - gamma * (1 - p) ** (gamma - 1) * np.log(p))
    g0 = -(1 - alpha) * (gamma * p ** (gamma - 1) * np.log(1 - p) - p ** gamma / (1 - p))
    return np.where(inside, np.where(t == 1, g1, np.where(t == 0, g0, 0.0)), 0.0)
