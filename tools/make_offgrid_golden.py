"""Known answers of the prebuilt pc_error_d for a decoded cloud that is NOT on the integer grid (a rate section with
scale != 1: the reference scales the reconstruction back by 1 / scale as float32 and writes the fractions to the ply,
process.py:70-78) -> tests/golden/pc_error_offgrid.npz.  Build container only (runs /root/reference/myutils/pc_error_d and
the reference's own write_ply_data).

    python tools/make_offgrid_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
KEYS = ["mse1      (p2point)", "mse2      (p2point)", "mseF      (p2point)", "mseF,PSNR (p2point)",
        "mse1      (p2plane)", "mse2      (p2plane)", "mseF      (p2plane)", "mseF,PSNR (p2plane)",
        "h.       1(p2point)", "h.       2(p2point)"]


def main():
    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    from pcgcv1_amd import synthetic
    os.chdir(REF)
    sys.path.insert(0, REF)
    from dataprocess import inout_points as rp
    tmp = tempfile.mkdtemp(prefix="offgrid_")
    out = {"keys": np.array(KEYS)}
    cases = [(7, 96, 0.625), (8, 96, 0.75), (9, 64, 0.3)]          # small clouds: the fixture stays under 1 MB
    for i, (seed, res, scale) in enumerate(cases):
        a = synthetic.make_cloud(seed=seed, res=res, n_shells=2, rmin=0.2, rmax=0.45).astype(np.int32)
        rng = np.random.default_rng(seed)
        a = a[np.sort(rng.choice(len(a), min(len(a), 30000), replace=False))]
        c = a.mean(0)
        na = (a - c) / np.maximum(np.linalg.norm(a - c, axis=1, keepdims=True), 1e-9)
        na = np.round(na + 0.05 * rng.standard_normal(na.shape), 6).astype(np.float32)
        down = np.unique(np.round(a.astype("float32") * scale), axis=0).astype(np.int32)
        keep = rng.random(len(down)) < 0.9                   # a lossy reconstruction: points dropped, a few moved
        moved = down[~keep][:500] + rng.integers(-1, 2, (min(500, int((~keep).sum())), 3))
        down = np.unique(np.concatenate([down[keep], moved]), axis=0)
        b = down.astype("float32") * float(1 / scale)       # process.py:76-77
        fa, fb = os.path.join(tmp, "a%d.ply" % i), os.path.join(tmp, "b%d.ply" % i)
        with open(fa, "w") as fh:
            fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                     "property float nx\nproperty float ny\nproperty float nz\nend_header\n" % len(a))
            np.savetxt(fh, np.concatenate([a.astype(np.float64), na], 1), fmt="%d %d %d %.6f %.6f %.6f")
        rp.write_ply_data(fb, b)
        txt = subprocess.run(["myutils/pc_error_d", "-a", fa, "-b", fb, "-n", fa, "--hausdorff=1", "-r", str(res - 1)],
                             capture_output=True, text=True).stdout
        vals = {}
        for line in txt.splitlines():
            for key in KEYS:
                if line.strip().startswith(key):
                    vals[key] = float(line.split(":")[-1])
        assert sorted(vals) == sorted(KEYS), txt
        out["a%d" % i], out["na%d" % i], out["b%d" % i] = a, na, b
        out["res%d" % i], out["scale%d" % i] = np.array(res), np.array(scale)
        out["vals%d" % i] = np.array([vals[k] for k in KEYS])
        print(i, len(a), len(b), vals)
    out["n_cases"] = np.array(len(cases))
    dst = os.path.join(ROOT, "tests", "golden", "pc_error_offgrid.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
