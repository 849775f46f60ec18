"""Soak test of the pipelined codec (GPU box): N steps of compress_hyper + decompress_hyper on the bench workload, every
step's strings and reconstruction compared bit for bit with the first step's (races between the two host pipelines,
the worker pools or the streams would show up as a difference or a decode error).  The steps follow each other WITHOUT a
device synchronisation in windows of `window` steps (as bench.py's timed loop does), the comparison happens at the end of a
window.   python tools/soak.py [steps] [profile: sparse | mid | dense | trained] [window]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from pcgcv1_amd import checkpoint, process, synthetic, transform   # noqa: E402
from pcgcv1_amd.models import model_voxception as model             # noqa: E402


def main(steps=100, profile="sparse", window=1):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    checkpoint._CACHE["soak"] = (checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00")) if profile == "trained"
                                 else synthetic.make_weights(seed=1300, profile=profile))
    cubes, _, _ = process.preprocess_points(synthetic.make_cloud(seed=1300), 1.0, 64, 64)
    ref, pending = None, []

    def check(it, out, xs):
        assert list(out[0]) == list(ref[0][0]), "y strings differ at step %d" % it
        assert bytes(out[4]) == bytes(ref[0][4]), "z string differs at step %d" % it
        for a, b in zip(out[1:4], ref[0][1:4]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), "ranges differ at step %d" % it
        assert torch.equal(xs, ref[1]), "reconstruction differs at step %d" % it
    for it in range(steps):
        out = transform.compress_hyper(cubes, model, "soak")
        xs = transform.decompress_hyper(*out, model, "soak")
        if ref is None:
            torch.cuda.synchronize()
            ref = (out, xs.clone())
            continue
        pending.append((it, out, xs))
        if len(pending) >= window or it + 1 == steps:
            torch.cuda.synchronize()
            for p in pending:
                check(*p)
            pending = []
    print("soak ok: %d steps of %d cubes (%s, windows of %d steps without synchronisation), every step bit-identical to the first"
          % (steps, int(cubes.shape[0]), profile, window))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, sys.argv[2] if len(sys.argv) > 2 else "sparse",
         int(sys.argv[3]) if len(sys.argv) > 3 else 1)
