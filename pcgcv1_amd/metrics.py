"""Rate / distortion figures of the reference's eval path (eval.py:102-111 bpp, 194-207 D1 via myutils/pc_error_d).

`d1_psnr(a, b, resolution)` reproduces what MPEG pc_error 0.13.4 prints as "mseF,PSNR (p2point)" for two
voxelised clouds: mse1 = mean_{p in A} min_{q in B} |p-q|^2, mse2 the same B->A, mseF = max, PSNR =
10 log10(3 peak^2 / mseF).  The nearest-neighbour search runs on the device (pcgc_d1_mse); the prebuilt
pc_error_d binary cannot ship, its answers on seeded clouds are pinned in tests/golden/pc_error_d1.npz.
"""
import numpy as np
import torch

from . import _lib


def _d1_one_way(a_d, b_d, res, ws):
    out = torch.empty(2, dtype=torch.float64, device=a_d.device)
    _lib.check(_lib.hip().pcgc_d1_mse(_lib.dptr(a_d), a_d.shape[0], _lib.dptr(b_d), b_d.shape[0], res, _lib.dptr(out),
                                      _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_d1_mse")
    mse, h2 = (float(v) for v in out.cpu().numpy())
    return mse, h2


def d1_metrics(points_a, points_b, resolution):
    """resolution = peak value (pc_error -r), e.g. 1023 for a 10-bit cloud.  Returns a dict with the pc_error keys."""
    dev = _lib.require_gpu()
    res = int(max(int(np.max(points_a)), int(np.max(points_b))) + 1)
    a_d = torch.from_numpy(np.ascontiguousarray(points_a, np.int32)).to(dev)
    b_d = torch.from_numpy(np.ascontiguousarray(points_b, np.int32)).to(dev)
    ws = torch.empty(int(_lib.hip().pcgc_d1_workspace_bytes(res)), dtype=torch.uint8, device=dev)
    mse1, h1 = _d1_one_way(a_d, b_d, res, ws)
    mse2, h2 = _d1_one_way(b_d, a_d, res, ws)
    mse_f = max(mse1, mse2)
    peak = float(resolution)

    def psnr(m):
        return float("inf") if m == 0 else 10.0 * np.log10(3.0 * peak * peak / m)
    return {"mse1      (p2point)": mse1, "mse2      (p2point)": mse2, "mseF      (p2point)": mse_f,
            "mse1,PSNR (p2point)": psnr(mse1), "mse2,PSNR (p2point)": psnr(mse2), "mseF,PSNR (p2point)": psnr(mse_f),
            "h.       1(p2point)": h1, "h.       2(p2point)": h2, "h.        (p2point)": max(h1, h2)}


def d1_psnr(points_a, points_b, resolution):
    return d1_metrics(points_a, points_b, resolution)["mseF,PSNR (p2point)"]


def bpp(total_bytes, n_input_points):
    """eval.py:102-111: 8 * bytes of all files / points of the ORIGINAL cloud."""
    return 8.0 * float(total_bytes) / float(n_input_points)
