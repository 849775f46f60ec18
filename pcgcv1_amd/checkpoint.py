"""Weight loading for the operator classes (replaces tf.train.Checkpoint(...).restore(
tf.train.latest_checkpoint(ckpt_dir)), transform.py:107-112, 214-218).

`ckpt_dir` may be
  * "synthetic", "synthetic:<seed>", "synthetic:<seed>:<profile>"  — seeded random weights of the
    reference's architecture (pcgcv1_amd/synthetic.py; there are no real checkpoints offline);
  * a directory holding `weights.npz` (or a path to an .npz) whose keys are the reference's
    checkpoint variable paths, e.g. "analysis_transform/vrn1_1/conv1_1/kernel",
    "estimator/bais_0", arrays in TensorFlow layouts.
Reading TensorFlow's own tensor-bundle files (ckpt-N.index / .data-*) is a "next" row of
SURVEY.md §8f and is not implemented yet; `tools/` will grow a converter.
"""
import os

import numpy as np

from . import synthetic

_CACHE = {}


def load(ckpt_dir):
    key = str(ckpt_dir)
    if key in _CACHE:
        return _CACHE[key]
    if key == "" or key.startswith("synthetic"):
        parts = key.split(":")
        seed = int(parts[1]) if len(parts) > 1 and parts[1] else 1300
        profile = parts[2] if len(parts) > 2 else "sparse"
        w = synthetic.make_weights(seed=seed, profile=profile)
    else:
        path = key
        if os.path.isdir(path):
            path = os.path.join(path, "weights.npz")
        if not os.path.exists(path):
            raise FileNotFoundError("no weights.npz under %r (TF tensor-bundle checkpoints are not readable yet; "
                                    "use 'synthetic[:seed[:profile]]' for seeded weights)" % key)
        with np.load(path) as z:
            w = {k: z[k] for k in z.files}
    _CACHE[key] = w
    return w


def save(weights, ckpt_dir):
    os.makedirs(ckpt_dir, exist_ok=True)
    np.savez(os.path.join(ckpt_dir, "weights.npz"), **weights)
