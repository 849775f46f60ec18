"""loss.py of the reference on MI355X: get_bce_loss (loss.py:8-33), get_confusion_matrix (35-58),
get_classify_metrics (60-78), get_focal_loss (83-93) — same names, arguments and return arity, torch tensors
instead of TF tensors.  Every reduction runs in libpcgc_hip.so (tail.hip: wavefront ballot / butterfly reductions,
fixed-order two-stage sums, so a value does not depend on the launch geometry); there is no CPU fallback."""
import torch

from . import _lib


def _pair(a, b):
    dev = _lib.require_gpu()
    a = a if torch.is_tensor(a) else torch.as_tensor(a)
    b = b if torch.is_tensor(b) else torch.as_tensor(b)
    a, b = a.to(dev, torch.float32).contiguous(), b.to(dev, torch.float32).contiguous()
    if a.numel() != b.numel():
        raise ValueError("pred and label differ in size: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    return a, b, dev


def get_bce_loss(pred, label):
    """(Weighted) binary cross entropy (loss.py:8-33): pred are logits, label is 0 / 1.
    Returns (empty_loss, full_loss) = mean over label == 0 of -log(1 - o), mean over label > 0 of -log(o),
    o = clip(sigmoid(pred), 1e-7, 1 - 1e-7).  An empty class gives nan, like tf.reduce_mean of nothing."""
    pred, label, dev = _pair(pred, label)
    lib = _lib.hip()
    sums = torch.empty(4, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.pcgc_bce_workspace_bytes(pred.numel())), dtype=torch.uint8, device=dev)
    _lib.check(lib.pcgc_bce_sums(_lib.dptr(pred), _lib.dptr(label), pred.numel(), _lib.dptr(sums), _lib.dptr(ws), ws.numel(),
                                 _lib.stream()), "pcgc_bce_sums")
    s0, n0, s1, n1 = sums.cpu().tolist()
    return (s0 / n0 if n0 else float("nan")), (s1 / n1 if n1 else float("nan"))


def get_confusion_matrix(pred, label, th=0.):
    """loss.py:35-58: TP, FP, FN maps (float32, the trailing channel axis squeezed) of (pred > th) vs (label > th)."""
    pred, label, dev = _pair(pred, label)
    shape = tuple(pred.shape[:-1]) if pred.dim() > 1 and pred.shape[-1] == 1 else tuple(pred.shape)
    tp, fp, fn = (torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(3))
    _lib.check(_lib.hip().pcgc_confusion_matrix(_lib.dptr(pred), _lib.dptr(label), pred.numel(), float(th), _lib.dptr(tp),
                                                _lib.dptr(fp), _lib.dptr(fn), _lib.stream()), "pcgc_confusion_matrix")
    return tp, fp, fn


def classify_counts(pred, label, th=0.):
    """(TP, FP, FN) as exact counts, without materialising the maps."""
    pred, label, dev = _pair(pred, label)
    lib = _lib.hip()
    sums = torch.empty(3, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.pcgc_classify_workspace_bytes()), dtype=torch.uint8, device=dev)
    _lib.check(lib.pcgc_classify_sums(_lib.dptr(pred), _lib.dptr(label), pred.numel(), float(th), _lib.dptr(sums), _lib.dptr(ws),
                                      ws.numel(), _lib.stream()), "pcgc_classify_sums")
    return tuple(sums.cpu().tolist())


def get_classify_metrics(pred, label, th=0.):
    """loss.py:60-78: (precision, recall, IoU); 0/0 is nan as in the reference's float division."""
    tp, fp, fn = classify_counts(pred, label, th)

    def div(a, b):
        return a / b if b else float("nan")
    return div(tp, tp + fp), div(tp, tp + fn), div(tp, tp + fp + fn)


def get_focal_loss(y_pred, y_true, gamma=2, alpha=0.9):
    """loss.py:83-93 (y_pred: probabilities, y_true: 0 / 1)."""
    y_pred, y_true, dev = _pair(y_pred, y_true)
    lib = _lib.hip()
    out = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.pcgc_focal_workspace_bytes()), dtype=torch.uint8, device=dev)
    _lib.check(lib.pcgc_focal_loss(_lib.dptr(y_pred), _lib.dptr(y_true), y_pred.numel(), float(gamma), float(alpha), _lib.dptr(out),
                                   _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_focal_loss")
    return float(out.cpu()[0])


def focal_loss_grad(y_pred, y_true, gamma=2, alpha=0.9, grad_scale=1.0):
    """d get_focal_loss / d y_pred (what a tf.GradientTape over loss.py:83-93 gives), times grad_scale."""
    y_pred, y_true, dev = _pair(y_pred, y_true)
    g = torch.empty_like(y_pred)
    _lib.check(_lib.hip().pcgc_focal_loss_bwd(_lib.dptr(y_pred), _lib.dptr(y_true), y_pred.numel(), float(gamma), float(alpha),
                                              float(grad_scale), _lib.dptr(g), _lib.stream()), "pcgc_focal_loss_bwd")
    return g
