"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

CPU restatement of the reference's hyperprior pipeline

  transform.py:91-197    compress_hyper    (order of operations, per-cube loops)
  transform.py:200-259   decompress_hyper
  loss.py:8-33           get_bce_loss
  loss.py:35-78          get_confusion_matrix / get_classify_metrics
  train_hyper.py:184-199 forward pass + rate terms of the training step

built from oracle/nets.py (conv stacks), oracle/entropy.py (entropy models) and
oracle/coder.c (range coder).  One cube per network call, like the reference's
tf.map_fn(parallel_iterations=1) — that is also what bench.py times as
`cpu_baseline` (kind "port": the literal reference needs TensorFlow 1.13, which
is absent).  *** PARITY UNPINNED *** for the learned operators (see nets.py).

`weights` is a flat dict keyed like the reference's tf.train.Checkpoint
(transform.py:107-111): analysis_transform/..., synthesis_transform/...,
hyper_encoder/..., hyper_decoder/..., estimator/{matrix_i,bais_i,factor_i}.
"""
import time

import numpy as np

from . import entropy, nets

LOWER_BOUND = np.float32(1e-9)      # transform.py:145, 232


def compress_hyper(cubes, weights, decompress=False, timers=None):
    t = timers if timers is not None else {}
    x = np.asarray(cubes, np.float32)
    wa, ws = nets.sub(weights, "analysis_transform"), nets.sub(weights, "synthesis_transform")
    whe, whd = nets.sub(weights, "hyper_encoder"), nets.sub(weights, "hyper_decoder")
    eb = nets.sub(weights, "estimator")

    t0 = time.time()
    ys = np.concatenate([nets.analysis_transform(wa, x[i:i + 1]) for i in range(len(x))])
    t["analysis"] = time.time() - t0
    t0 = time.time()
    zs = np.concatenate([nets.hyper_encoder(whe, ys[i:i + 1]) for i in range(len(x))])
    t["hyper_encoder"] = time.time() - t0
    z_hats, _ = entropy.eb_call(eb, zs, training=False)
    t0 = time.time()
    ls = [nets.hyper_decoder(whd, z_hats[i:i + 1]) for i in range(len(x))]
    locs = np.concatenate([l[0] for l in ls])
    scales = np.maximum(np.concatenate([l[1] for l in ls]), LOWER_BOUND)
    t["hyper_decoder"] = time.time() - t0

    t0 = time.time()
    z_string, z_min_v, z_max_v = entropy.eb_compress(eb, zs)
    z_shape = np.array(zs.shape, np.int32)
    t["entropy_encode_hyper"] = time.time() - t0

    t0 = time.time()
    y_strings, y_min_vs, y_max_vs = [], [], []
    for i in range(len(x)):
        s, mn, mx = entropy.sc_compress(ys[i:i + 1], locs[i:i + 1], scales[i:i + 1])
        y_strings.append(s)
        y_min_vs.append(mn)
        y_max_vs.append(mx)
    y_shape = np.array((1,) + ys.shape[1:], np.int32)
    t["entropy_encode"] = time.time() - t0
    out = (y_strings, np.array(y_min_vs, np.int32), np.array(y_max_vs, np.int32), y_shape,
           z_string, z_min_v, z_max_v, z_shape)
    if decompress:
        yd = np.concatenate([entropy.sc_decompress(y_strings[i], locs[i:i + 1], scales[i:i + 1],
                                                   y_min_vs[i], y_max_vs[i], y_shape) for i in range(len(x))])
        xd = np.concatenate([nets.synthesis_transform(ws, yd[i:i + 1]) for i in range(len(x))])
        return out + (xd,)
    return out


def decompress_hyper(y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape,
                     weights, timers=None):
    t = timers if timers is not None else {}
    ws = nets.sub(weights, "synthesis_transform")
    whd = nets.sub(weights, "hyper_decoder")
    eb = nets.sub(weights, "estimator")
    t0 = time.time()
    zs = entropy.eb_decompress(eb, z_string, z_min_v, z_max_v, z_shape)
    t["entropy_decode_hyper"] = time.time() - t0
    t0 = time.time()
    ls = [nets.hyper_decoder(whd, zs[i:i + 1]) for i in range(len(zs))]
    locs = np.concatenate([l[0] for l in ls])
    scales = np.maximum(np.concatenate([l[1] for l in ls]), LOWER_BOUND)
    t["hyper_decoder"] = time.time() - t0
    t0 = time.time()
    ys = np.concatenate([entropy.sc_decompress(y_strings[i], locs[i:i + 1], scales[i:i + 1],
                                               y_min_vs[i], y_max_vs[i], y_shape) for i in range(len(zs))])
    t["entropy_decode"] = time.time() - t0
    t0 = time.time()
    xs = np.concatenate([nets.synthesis_transform(ws, ys[i:i + 1]) for i in range(len(zs))])
    t["synthesis"] = time.time() - t0
    return xs


# ---------------------------------------------------------------------------
# loss.py
# ---------------------------------------------------------------------------
def bce_loss(pred, label):
    """loss.py:8-33 -> (empty_loss, full_loss), float32 arithmetic, float64 accumulation of the mean."""
    pred = np.asarray(pred, np.float32)
    occ = np.clip((1.0 / (1.0 + np.exp(-pred))).astype(np.float32), np.float32(1e-7), np.float32(1.0 - 1e-7))
    lab = np.asarray(label).max(axis=-1)
    occ = occ[..., 0]
    neg, pos = occ[lab == 0], occ[lab > 0]
    return (float(np.mean(-np.log(np.float32(1.0) - neg), dtype=np.float64)),
            float(np.mean(-np.log(pos), dtype=np.float64)))


def classify_metrics(pred, label, th=0.0):
    """loss.py:35-78 -> (precision, recall, IoU)."""
    p = (np.asarray(pred)[..., 0] > th).astype(np.float32)
    l = (np.asarray(label)[..., 0] > th).astype(np.float32)
    tp, fp, fn = (p * l).sum(), (p * (1 - l)).sum(), ((1 - p) * l).sum()
    return tp / (tp + fp), tp / (tp + fn), tp / (tp + fp + fn)


def rate_terms(weights, x, noise_y=None, noise_z=None, lower_bound=1e-9):
    """train_hyper.py:184-196 forward (eval mode when no noise is given):
    returns dict(bpp_y, bpp_z, x_tilde, y, z)."""
    wa, ws = nets.sub(weights, "analysis_transform"), nets.sub(weights, "synthesis_transform")
    whe, whd = nets.sub(weights, "hyper_encoder"), nets.sub(weights, "hyper_decoder")
    eb = nets.sub(weights, "estimator")
    x = np.asarray(x, np.float32)
    y = nets.analysis_transform(wa, x)
    z = nets.hyper_encoder(whe, y)
    z_t, lz = entropy.eb_call(eb, z, training=noise_z is not None, noise=noise_z)
    loc, scale = nets.hyper_decoder(whd, z_t)
    scale = np.maximum(scale, np.float32(lower_bound))
    y_t, ly = entropy.sc_call(y, loc, scale, training=noise_y is not None, noise=noise_y)
    x_t = nets.synthesis_transform(ws, y_t)
    num_points = float(x.sum())
    return dict(bpp_y=float(np.log(ly.astype(np.float64)).sum() / (-np.log(2.0) * num_points)),
                bpp_z=float(np.log(lz.astype(np.float64)).sum() / (-np.log(2.0) * num_points)),
                x_tilde=x_t, y=y, z=z, loc=loc, scale=scale)
