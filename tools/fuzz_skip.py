"""Randomised check of the exact empty-space skipping (GPU box): random batches of 64^3 occupancy cubes — surfaces, planes on
tile boundaries, lines, single voxels, dense blocks, empty cubes, batch sizes that leave ragged chunks — through
AnalysisTransform with PCGC_SKIP_EMPTY = 1 (virtual tiles), 2 (copies) and 3 (blocks on slots of 16 voxels) against 0 (everything computed): the latents must
be bit-identical, with the workspace poisoned (NaN) before every run.  Also the two checkpoints' worth of weights.
    python tools/fuzz_skip.py [batches] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from pcgcv1_amd import transform   # noqa: E402
from pcgcv1_amd.models import model_voxception as model   # noqa: E402


def random_cube(rng):
    x = np.zeros((64, 64, 64), np.float32)
    kind = rng.integers(0, 9)
    if kind == 0:                                            # empty
        pass
    elif kind == 1:                                          # a few single voxels, often on tile / cube borders
        for _ in range(rng.integers(1, 6)):
            p = [int(rng.choice([0, 1, 7, 8, 15, 16, 31, 32, 62, 63, rng.integers(0, 64)])) for _ in range(3)]
            x[p[0], p[1], p[2]] = 1
    elif kind == 2:                                          # an axis-aligned plane
        a, k = rng.integers(0, 3), int(rng.choice([0, 7, 8, 9, 31, 32, 56, 63, rng.integers(0, 64)]))
        idx = [slice(None)] * 3
        idx[a] = k
        x[tuple(idx)] = 1
    elif kind == 3:                                          # a sphere shell
        c, r = rng.uniform(-10, 74, 3), rng.uniform(5, 60)
        g = np.stack(np.meshgrid(*[np.arange(64)] * 3, indexing="ij"), -1)
        d = np.linalg.norm(g - c, axis=-1)
        x[np.abs(d - r) < 0.7] = 1
    elif kind == 4:                                          # a tilted plane
        n = rng.standard_normal(3)
        n /= np.linalg.norm(n)
        g = np.stack(np.meshgrid(*[np.arange(64)] * 3, indexing="ij"), -1)
        x[np.abs((g - rng.uniform(0, 64, 3)) @ n) < 0.6] = 1
    elif kind == 5:                                          # lines along each axis
        for _ in range(rng.integers(1, 4)):
            a = rng.integers(0, 3)
            idx = [int(rng.integers(0, 64)) for _ in range(3)]
            idx[a] = slice(None)
            x[tuple(idx)] = 1
    elif kind == 6:                                          # a dense block
        lo = rng.integers(0, 56, 3)
        hi = lo + rng.integers(1, 9, 3)
        x[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = 1
    elif kind == 7:                                          # random sparse noise
        x[rng.random((64, 64, 64)) < rng.choice([1e-5, 1e-4, 1e-3, 0.02])] = 1
    else:                                                    # a slab of noise in a few planes / rows only
        a = rng.integers(0, 3)
        k = rng.integers(0, 60)
        idx = [slice(None)] * 3
        idx[a] = slice(k, k + int(rng.integers(1, 4)))
        sub = x[tuple(idx)]
        sub[rng.random(sub.shape) < 0.05] = 1
    return x


def main(batches=60, seed=0):
    rng = np.random.default_rng(seed)
    nets = [transform.get_codec(model, "synthetic:77:dense").analysis_transform,
            transform.get_codec(model, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "checkpoints", "hyper",
                                                    "a6.00b3.00")).analysis_transform]
    cubes_total = 0
    for it in range(batches):
        B = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 24, 31, 33, 40, 47, 65, 103]))
        x = torch.from_numpy(np.stack([random_cube(rng) for _ in range(B)])[..., None]).cuda()
        net = nets[it % 2]
        os.environ["PCGC_SKIP_EMPTY"] = "0"
        y0 = net(x).clone()
        for mode in ("1", "2", "3"):
            os.environ["PCGC_SKIP_EMPTY"] = mode
            for ws in net._ws.values():
                ws.fill_(255)
            y = net(x)
            if not torch.equal(y, y0):
                bad = (y != y0).reshape(B, -1).any(dim=1).nonzero().reshape(-1).tolist()
                raise SystemExit("MISMATCH: batch %d (B = %d, seed %d), mode %s, cubes %r" % (it, B, seed, mode, bad))
        cubes_total += B
    os.environ.pop("PCGC_SKIP_EMPTY", None)
    print("fuzz ok: %d random batches, %d cubes, modes 1, 2 and 3 bit-identical to 0 (seed %d)" % (batches, cubes_total, seed))


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:3]))
