"""Weight loading for the operator classes (replaces tf.train.Checkpoint(...).restore(
tf.train.latest_checkpoint(ckpt_dir)), transform.py:107-112, 214-218).

`ckpt_dir` may be
  * "synthetic", "synthetic:<seed>", "synthetic:<seed>:<profile>"  — seeded random weights of the
    reference's architecture (pcgcv1_amd/synthetic.py; there are no real checkpoints offline); profile
    "simple" gives models/model_simple.py weights for --mode=factorized --modelname=models.model_simple;
  * a directory holding `weights.npz` (or a path to an .npz) whose keys are the reference's
    checkpoint variable paths, e.g. "analysis_transform/vrn1_1/conv1_1/kernel",
    "estimator/bais_0", arrays in TensorFlow layouts.
  * a TensorFlow checkpoint directory (`checkpoint` state file and/or ckpt-N.index + .data-00000-of-00001),
    read by pcgcv1_amd/tf_bundle.py without TensorFlow; optimizer slots, global_step and the serialized
    object graph are ignored.
`save(weights, dir)` writes weights.npz; `save_tf(weights, dir, step)` writes a tensor bundle the way
train_hyper.py:255-268 does.
"""
import os

import numpy as np

from . import synthetic, tf_bundle

_PREFIXES = ("analysis_transform/", "synthesis_transform/", "hyper_encoder/", "hyper_decoder/", "estimator/")


def _from_bundle(prefix):
    raw = tf_bundle.read_bundle(prefix)
    w = {k: v.astype(np.float32) for k, v in raw.items()
         if k.startswith(_PREFIXES) and ".OPTIMIZER_SLOT" not in k and v.dtype.kind == "f"}
    if not w:
        raise ValueError("%s: no model variables (analysis_transform/..., estimator/...) among %d tensors; first keys: %s"
                         % (prefix, len(raw), sorted(raw)[:5]))
    return w

class _Cache(dict):
    """ckpt_dir -> weights.  Keys set from outside (tests, bench.py register seeded weights under a name) are
    remembered as synthetic: only those may fall back to initialiser-built parts (transform._entropy_bottleneck_y)."""

    def __setitem__(self, key, value):
        if not _LOADING:
            _CACHE_SYNTHETIC.add(key)
        dict.__setitem__(self, key, value)


_LOADING = False
_CACHE_SYNTHETIC = set()
_CACHE = _Cache()


def load(ckpt_dir):
    key = str(ckpt_dir)
    if key in _CACHE:
        return _CACHE[key]
    if key == "" or key.startswith("synthetic"):
        parts = key.split(":")
        seed = int(parts[1]) if len(parts) > 1 and parts[1] else 1300
        profile = parts[2] if len(parts) > 2 else "sparse"
        # profile 'simple' = the factorized ablation model (models/model_simple.py + 32-channel bottleneck)
        w = synthetic.make_weights_simple(seed=seed) if profile == "simple" else synthetic.make_weights(seed=seed, profile=profile)
    else:
        path = key
        bundle = None
        if os.path.isdir(path):
            if os.path.exists(os.path.join(path, "weights.npz")):
                path = os.path.join(path, "weights.npz")
            else:
                bundle = tf_bundle.latest_checkpoint(path)
        elif os.path.exists(path + ".index"):
            bundle = path
        if bundle:
            w = _from_bundle(bundle)
        elif os.path.isfile(path):
            with np.load(path) as z:
                w = {k: z[k] for k in z.files}
        else:
            raise FileNotFoundError("%r holds neither weights.npz nor a TensorFlow checkpoint (checkpoint / ckpt-N.index); "
                                    "use 'synthetic[:seed[:profile]]' for seeded weights" % key)
    global _LOADING
    _LOADING = True
    try:
        _CACHE[key] = w
    finally:
        _LOADING = False
    return w


def save(weights, ckpt_dir):
    os.makedirs(ckpt_dir, exist_ok=True)
    np.savez(os.path.join(ckpt_dir, "weights.npz"), **weights)


def save_tf(weights, ckpt_dir, step):
    """Tensor-bundle checkpoint ckpt-<step> + `checkpoint` state file (train_hyper.py:255-268)."""
    _CACHE.pop(str(ckpt_dir), None)
    return tf_bundle.save_checkpoint(ckpt_dir, step, weights)
