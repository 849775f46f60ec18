"""Host-side count: what fraction of the analysis' wave tiles would have to be COMPUTED (their receptive field holds an occupied voxel)
for other tile shapes than the full 64-voxel rows the 64^3 row kernels work on?  Bench cloud, a sample of cubes, a window dilated by
radius r (r = 2 ... 8 over the three C = 16 blocks).  No GPU needed.
    python tools/exp/count_tiles.py [n_cubes=60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pcgcv1_amd import synthetic

n_cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pts = np.asarray(synthetic.make_cloud(seed=1300), np.int64)
uk, inv = np.unique(pts // 64, axis=0, return_inverse=True)
keep = np.nonzero(np.bincount(inv) >= 64)[0][:n_cubes]
occs = []
for k in keep:
    p = pts[inv == k] % 64
    o = np.zeros((64, 64, 64), np.int32)
    o[p[:, 0], p[:, 1], p[:, 2]] = 1
    occs.append(o)


def dilate(o, r):
    d = o
    for ax in range(3):
        c = np.cumsum(np.concatenate([np.zeros_like(np.take(d, [0], axis=ax)), d], axis=ax), axis=ax)
        n = d.shape[ax]
        lo, hi = np.clip(np.arange(n) - r, 0, n), np.clip(np.arange(n) + r + 1, 0, n)
        d = ((np.take(c, hi, axis=ax) - np.take(c, lo, axis=ax)) > 0).astype(np.int32)
    return d


tiles = ((8, 2, 64), (1, 1, 64), (8, 4, 16), (4, 4, 16), (8, 8, 8), (2, 8, 8), (1, 1, 1))
print("%d cubes of the bench cloud, mean occupancy %.4f; tile = planes x rows x voxels along the row" % (len(occs), np.mean([o.mean() for o in occs])))
print("radius  " + "  ".join("%-10s" % ("%dx%dx%d" % t) for t in tiles))
for r in (2, 4, 6, 8):
    row = []
    for td, th, tw in tiles:
        heavy = tot = 0
        for o in occs:
            t = dilate(o, r).reshape(64 // td, td, 64 // th, th, 64 // tw, tw).max(axis=(1, 3, 5))
            heavy += int(t.sum())
            tot += t.size
        row.append(heavy / tot)
    print("%6d  " % r + "  ".join("%-10.3f" % v for v in row))
