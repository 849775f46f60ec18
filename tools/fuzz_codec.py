"""Randomised round trips of the hyperprior codec (GPU box): random batches of random geometry (tools/fuzz_skip.random_cube)
and sizes that cross the pipeline / slice / chunk boundaries (1 ... 210 cubes).  For every batch
  * the pipelined compress_hyper (two host pipelines from 96 cubes) gives the same strings and ranges as the staged path
    (compress_hyper(..., decompress=True): one pipeline, the reference's stage order),
  * decompress_hyper of those strings is bit-identical to the encoder-side reconstruction of the staged path,
  * a second decode is bit-identical to the first.
    python tools/fuzz_codec.py [batches] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from fuzz_skip import random_cube   # noqa: E402
from pcgcv1_amd import transform   # noqa: E402
from pcgcv1_amd.models import model_voxception as model   # noqa: E402


def main(batches=40, seed=0):
    rng = np.random.default_rng(seed)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ckpts = [os.path.join(root, "checkpoints", "hyper", r) for r in ("a0.75b3.00", "a6.00b3.00", "a16.00b3.00")] + ["synthetic:7:sparse"]
    total = 0
    for it in range(batches):
        B = int(rng.choice([1, 3, 8, 17, 40, 95, 96, 97, 103, 128, 150, 205, 210]))
        x = torch.from_numpy(np.stack([random_cube(rng) for _ in range(B)])[..., None]).cuda()
        ck = ckpts[it % len(ckpts)]
        staged = transform.compress_hyper(x, model, ck, decompress=True)
        out = transform.compress_hyper(x, model, ck)
        for k in range(8):
            a, b = staged[k], out[k]
            same = (list(a) == list(b)) if k == 0 else (bytes(a) == bytes(b) if isinstance(a, (bytes, bytearray, memoryview)) else
                                                        np.array_equal(np.asarray(a), np.asarray(b)))
            if not same:
                raise SystemExit("MISMATCH: batch %d (B = %d, %s, seed %d): output %d of compress_hyper differs between the paths" % (it, B, ck, seed, k))
        xs = transform.decompress_hyper(*out, model, ck)
        if not torch.equal(xs, staged[8]):
            raise SystemExit("MISMATCH: batch %d (B = %d, %s, seed %d): decoder != encoder-side reconstruction" % (it, B, ck, seed))
        if not torch.equal(transform.decompress_hyper(*out, model, ck), xs):
            raise SystemExit("MISMATCH: batch %d (B = %d, %s, seed %d): second decode differs" % (it, B, ck, seed))
        total += B
    print("codec fuzz ok: %d random batches, %d cubes: pipelined == staged strings, decoder == encoder-side reconstruction (seed %d)"
          % (batches, total, seed))


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:3]))
