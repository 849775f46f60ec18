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


_EB_LISTS = {"matrix": "_matrices", "bais": "_biases", "factor": "_factors"}      # entropy_model.py:47-66


def _expected_variables(present):
    """The variables the reference's tf.train.Checkpoint(...) root edges lead to (transform.py:107-111, 35-38), as
    (attribute path, alias paths).  `present(path_prefix)` says whether the file knows anything under a prefix: the
    architecture (model_voxception or model_simple) and the optional hyperprior parts are taken from the file."""
    from .models import spec
    simple = present("analysis_transform/conv_1") and not present("analysis_transform/conv_in")
    nets = dict(spec.SIMPLE_NETS) if simple else {k: v for k, v in spec.NETS.items()
                                                  if k in ("analysis_transform", "synthesis_transform") or present(k)}
    out = []
    for net, layers in nets.items():
        for l in layers():
            out.append(("%s/%s/kernel" % (net, l.name), []))
            if l.bias:
                out.append(("%s/%s/bias" % (net, l.name), []))
    for i in range(4):
        for k, lst in _EB_LISTS.items():
            # Layer.add_variable tracks the variable under its name; the list attributes it is appended to are tracked
            # data structures too, so the same node is also estimator/_matrices/<i> (either edge may be what a file holds)
            out.append(("estimator/%s_%d" % (k, i), ["estimator/%s/%d" % (lst, i)]))
    return out


def _from_bundle(prefix):
    """Bind the model variables of a tensor bundle the way tf.train.Checkpoint.restore does (transform.py:107-112): walk the
    file's serialized object graph from the root along the attribute names of the reference's objects and take each
    variable node's checkpoint key; a file without a graph (tf.train.Saver style) or a variable the graph does not lead to
    falls back to the key name.  Every expected variable that could not be bound is reported."""
    raw = tf_bundle.read_bundle(prefix, strip=False)
    blob = tf_bundle.read_string_scalar(prefix)
    by_path = tf_bundle.graph_variables(tf_bundle.parse_object_graph(blob)) if blob else {}
    suffix = tf_bundle._SUFFIX
    names = set(by_path) | {k[:-len(suffix)] if k.endswith(suffix) else k for k in raw}

    def present(pfx):
        return any(n == pfx or n.startswith(pfx + "/") for n in names)
    w, missing, how = {}, [], {}
    for path, aliases in _expected_variables(present):
        arr = None
        for cand in [path] + aliases:                         # 1. the object graph, edge by edge
            key = by_path.get(cand)
            if key is not None and key in raw:
                arr, how[path] = raw[key], "graph:" + cand
                break
        if arr is None:                                       # 2. the key name
            for cand in [path] + aliases:
                for key in (cand + suffix, cand):
                    if key in raw:
                        arr, how[path] = raw[key], "key:" + key
                        break
                if arr is not None:
                    break
        if arr is None:
            missing.append(path)
        elif arr.dtype.kind != "f":
            raise ValueError("%s: variable %s is %s, expected a float tensor" % (prefix, path, arr.dtype))
        else:
            w[path] = arr.astype(np.float32)
    if not w:
        raise ValueError("%s: no model variables (analysis_transform/..., estimator/...) among %d tensors; first keys: %s"
                         % (prefix, len(raw), sorted(raw)[:5]))
    if missing:
        raise ValueError("%s: %d of %d expected variables could not be bound through the object graph or by key name: %s"
                         % (prefix, len(missing), len(missing) + len(w), ", ".join(missing[:12]) + (" ..." if len(missing) > 12 else "")))
    global LAST_BINDING
    LAST_BINDING = how
    return w


LAST_BINDING = {}      # variable path -> "graph:<path walked>" / "key:<checkpoint key>" of the last bundle read (diagnostics, tests)


def load_prefix(prefix):
    """The model variables of ONE bundle (`.../ckpt-N`), bound like load() binds them; not cached (the trainer resumes from
    files that are being rewritten)."""
    return _from_bundle(prefix)


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
