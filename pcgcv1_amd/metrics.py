"""Rate / distortion figures of the reference's eval path (eval.py:102-111 bpp, 194-207 D1 via myutils/pc_error_d).

`d1_psnr(a, b, resolution)` reproduces what MPEG pc_error 0.13.4 prints as "mseF,PSNR (p2point)" for two
voxelised clouds: mse1 = mean_{p in A} min_{q in B} |p-q|^2, mse2 the same B->A, mseF = max, PSNR =
10 log10(3 peak^2 / mseF).  The nearest-neighbour search runs on the device (pcgc_d1_mse); the prebuilt
pc_error_d binary cannot ship, its answers on seeded clouds are pinned in tests/golden/pc_error_d1.npz.
`d2_metrics` adds the point-to-plane figures ("mseF,PSNR (p2plane)", eval.py's D2) with pc_error's tie handling
and normal transfer (csrc/tail.hip), pinned in tests/golden/pc_error_d2.npz.
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


def _sorted_keys(points_d, res):
    p = points_d.to(torch.int64)
    keys = (p[:, 0] * res + p[:, 1]) * res + p[:, 2]
    keys, order = torch.sort(keys)
    if keys.numel() > 1 and bool((keys[1:] == keys[:-1]).any()):
        raise ValueError("point cloud holds duplicate points (pc_error drops them first; pass unique voxels)")
    return keys.contiguous(), order


def d2_metrics(points_a, normals_a, points_b, resolution):
    """Point-to-plane (D2) figures of `pc_error -a A -b B -n A` (myutils/pc_error_wrapper.py:46-51): A carries
    normals, B (the decoded cloud) receives them from A.  Returns the p2plane keys of the wrapper's table."""
    dev = _lib.require_gpu()
    lib = _lib.hip()
    res = int(max(int(np.max(points_a)), int(np.max(points_b))) + 1)
    a_d = torch.from_numpy(np.ascontiguousarray(points_a, np.int32)).to(dev)
    b_d = torch.from_numpy(np.ascontiguousarray(points_b, np.int32)).to(dev)
    na_d = torch.from_numpy(np.ascontiguousarray(normals_a, np.float32)).to(dev)
    ka, oa = _sorted_keys(a_d, res)
    kb, ob = _sorted_keys(b_d, res)
    a_s, na_s, b_s = a_d[oa].contiguous(), na_d[oa].contiguous(), b_d[ob].contiguous()
    ws = torch.empty(int(lib.pcgc_d2_workspace_bytes(res, max(len(ka), len(kb)))), dtype=torch.uint8, device=dev)
    nb_s = torch.empty((len(kb), 3), dtype=torch.float32, device=dev)
    _lib.check(lib.pcgc_d2_transfer_normals(_lib.dptr(a_s), len(ka), _lib.dptr(na_s), _lib.dptr(kb), len(kb), res, _lib.dptr(nb_s),
                                            _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_d2_transfer_normals")
    out = torch.empty(4, dtype=torch.float64, device=dev)
    _lib.check(lib.pcgc_d2_mse(_lib.dptr(a_s), len(ka), _lib.dptr(kb), len(kb), _lib.dptr(nb_s), res, _lib.dptr(out),
                               _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_d2_mse A->B")
    _lib.check(lib.pcgc_d2_mse(_lib.dptr(b_s), len(kb), _lib.dptr(ka), len(ka), _lib.dptr(na_s), res, _lib.dptr(out[2:]),
                               _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_d2_mse B->A")
    mse1, h1, mse2, h2 = (float(v) for v in out.cpu().numpy())
    mse_f, peak = max(mse1, mse2), float(resolution)

    def psnr(m):
        return float("inf") if m == 0 else 10.0 * np.log10(3.0 * peak * peak / m)
    return {"mse1      (p2plane)": mse1, "mse2      (p2plane)": mse2, "mseF      (p2plane)": mse_f,
            "mse1,PSNR (p2plane)": psnr(mse1), "mse2,PSNR (p2plane)": psnr(mse2), "mseF,PSNR (p2plane)": psnr(mse_f),
            "h.       1(p2plane)": h1, "h.       2(p2plane)": h2, "h.        (p2plane)": max(h1, h2)}


def _off_grid(points):
    p = np.asarray(points)
    return np.issubdtype(p.dtype, np.floating) and bool((p != np.rint(p)).any())


def pc_error_off_grid(points_a, points_b, normals_a=None, resolution=1023):
    """The same figures when the decoded cloud B is NOT on the integer grid: a rate point with scale != 1 (R1 = 5/8) is
    scaled back by 1 / scale as float32 and written as text (process.py:76-77), and pc_error measures those coordinates —
    rounding them first moves D1 by a whole dB.  The device kernels search a voxel bitmap and cannot hold such a cloud, so
    this path is a k-d tree on the host (scipy, float64 like pc_error): the eval harness' metric of ONE rate section, not
    part of the codec.  Same definitions as d1_metrics / d2_metrics: T(p) = every target point at the minimal distance,
    B's normals transferred from A as the plain mean over the points that chose it.  Pinned against the prebuilt pc_error_d
    on the config-3 frame's R1 section (tests/golden/oracle_a0.75b3_cloud2000_s0.625.npz)."""
    from scipy.spatial import cKDTree
    a = np.asarray(points_a, np.float64)
    b = np.unique(np.asarray(points_b, np.float32), axis=0).astype(np.float64)          # pc_error drops duplicate points
    peak = float(resolution)

    def psnr(m):
        return float("inf") if m == 0 else 10.0 * np.log10(3.0 * peak * peak / m)

    def tied(src, tree, k=8):
        """-> (squared NN distance per source point, list of (source index, target index) pairs of every tied nearest neighbour)"""
        d, idx = tree.query(src, k=k)
        d2 = d * d
        # pc_error's tie rule, pinned on the goldens: squared distances within 1e-8 (absolute, float64) of the minimum
        # (scale 3/4: |2 - 1.3333334| and |2.6666667 - 2| differ by 1e-7 and are NOT tied; 5.3333335 / 6.6666665 around 6 are)
        tie = np.abs(d2 - d2[:, :1]) < 1e-8                   # the first column is the minimum
        tie &= np.isfinite(d2)
        rows = np.nonzero(tie)
        return d2[:, 0], rows[0], idx[rows]
    tree_a, tree_b = cKDTree(a), cKDTree(b)
    d2_ab, ia, jb = tied(a, tree_b)
    d2_ba, ib, ja = tied(b, tree_a)
    mse1, mse2 = float(d2_ab.mean()), float(d2_ba.mean())
    out = {"mse1      (p2point)": mse1, "mse2      (p2point)": mse2, "mseF      (p2point)": max(mse1, mse2),
           "mse1,PSNR (p2point)": psnr(mse1), "mse2,PSNR (p2point)": psnr(mse2), "mseF,PSNR (p2point)": psnr(max(mse1, mse2)),
           "h.       1(p2point)": float(d2_ab.max()), "h.       2(p2point)": float(d2_ba.max()),
           "h.        (p2point)": float(max(d2_ab.max(), d2_ba.max()))}
    if normals_a is not None:
        na = np.asarray(normals_a, np.float64)
        nb = np.zeros((len(b), 3))
        cnt = np.zeros(len(b))
        np.add.at(nb, jb, na[ia])
        np.add.at(cnt, jb, 1.0)
        nb[cnt > 0] /= cnt[cnt > 0, None]

        def plane(src, dst, n_dst, i_src, j_dst):
            e = ((src[i_src] - dst[j_dst]) * n_dst[j_dst]).sum(1) ** 2
            tot, c = np.zeros(len(src)), np.zeros(len(src))
            np.add.at(tot, i_src, e)
            np.add.at(c, i_src, 1.0)
            return tot / c
        p1, p2 = plane(a, b, nb, ia, jb), plane(b, a, na, ib, ja)
        m1, m2 = float(p1.mean()), float(p2.mean())
        out.update({"mse1      (p2plane)": m1, "mse2      (p2plane)": m2, "mseF      (p2plane)": max(m1, m2),
                    "mse1,PSNR (p2plane)": psnr(m1), "mse2,PSNR (p2plane)": psnr(m2), "mseF,PSNR (p2plane)": psnr(max(m1, m2)),
                    "h.       1(p2plane)": float(p1.max()), "h.       2(p2plane)": float(p2.max()),
                    "h.        (p2plane)": float(max(p1.max(), p2.max()))})
    return out


def pc_error(points_a, points_b, normals_a=None, resolution=1023):
    """All figures of myutils/pc_error_wrapper.pc_error (26-75) as one dict: D1 always, D2 when normals are given.
    points_b with fractional coordinates (a rate point with scale != 1): pc_error_off_grid."""
    if _off_grid(points_b):
        return pc_error_off_grid(points_a, points_b, normals_a, resolution)
    out = d1_metrics(points_a, points_b, resolution)
    if normals_a is not None:
        out.update(d2_metrics(points_a, normals_a, points_b, resolution))
    return out


def bpp(total_bytes, n_input_points):
    """eval.py:102-111: 8 * bytes of all files / points of the ORIGINAL cloud."""
    return 8.0 * float(total_bytes) / float(n_input_points)
