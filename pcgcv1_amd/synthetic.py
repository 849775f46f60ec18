"""Seeded synthetic inputs: weights of the reference's architecture and
voxelised surface clouds.

The reference's checkpoints (README.md:20-22, download links only) and its test
clouds (longdress_vox10_1300.ply ...) exist neither in this repository nor on
the GPU box and there is no network, so every measurement uses

* `make_weights(seed)`  — random fp32 weights of exactly the reference's layer
  shapes (models/spec.py), He-scaled, with per-layer gains chosen (offline, with
  the CPU oracle, tools/calibrate_weights.py) so that the latents look like a
  mid-rate operating point: y-hat within about [-8, 8], z-hat within about
  [-6, 6], Laplace scales within (0.05, 2) — the container format needs
  |min|,|max| <= 15 (inout_bitstream.py:95-96) and >= 2 symbols
  (entropy_model.py:192-193);
* `make_cloud(seed, res, target)` — a closed surface (union of ellipsoid
  shells) voxelised on a res^3 grid, about 1.6 % occupancy in the occupied 64^3
  cubes like longdress (demo.ipynb:123-124: 202 cubes, mean 4246 points).

Keys follow the reference's tf.train.Checkpoint naming (transform.py:107-111,
entropy_model.py:50-66, including the `bais_` spelling).
"""
import numpy as np

from .models import spec

# Output-layer gains found by tools/calibrate_weights.py (oracle run, seed 1300).
#  "sparse": resembles a trained checkpoint's operating point — almost every
#            y-hat is 0, symbols within about [-2, 2] (demo.ipynb:229 records
#            [-2, 2] for factorized a2b3), a few tens..hundreds of bytes per cube
#            (demo.ipynb:700: 7128 B / 202 cubes).  bench.py's workload.
#  "dense" : stress profile — y-hat within about [-8, 8], wide CDF rows; used by
#            the parity tests to exercise the entropy path.
PROFILES = {
    "sparse": {"gains": {"analysis_transform": {"conv_out": 0.028}, "synthesis_transform": {"deconv_out": 1.0},
                         "hyper_encoder": {"conv3": 1.4}, "hyper_decoder": {"conv4_1": 0.0027, "conv4_2": 0.2326}},
               "scale_bias": 0.10, "scale_bias_jitter": 0.01},
    # "mid": the dense profile's latents scaled by 0.4 — y-hat within about [-7, 7] on the bench cloud: the second
    #        operating point bench.py reports (a wider support than "sparse": longer CDF rows, more coder work)
    "mid": {"gains": {"analysis_transform": {"conv_out": 0.158}, "synthesis_transform": {"deconv_out": 1.0},
                      "hyper_encoder": {"conv3": 0.6191}, "hyper_decoder": {"conv4_1": 0.576, "conv4_2": 0.2474}},
            "scale_bias": 0.25, "scale_bias_jitter": 0.03},
    "dense": {"gains": {"analysis_transform": {"conv_out": 0.3954}, "synthesis_transform": {"deconv_out": 1.0},
                        "hyper_encoder": {"conv3": 0.6191}, "hyper_decoder": {"conv4_1": 1.4397, "conv4_2": 0.2474}},
              "scale_bias": 0.35, "scale_bias_jitter": 0.05},
}
_RES_GAIN = 0.5          # gain on the last conv of each VRN path keeps 9 residual blocks bounded


def _he(rng, shape, fan_in, gain=1.0):
    return (rng.standard_normal(shape) * (gain * np.sqrt(2.0 / fan_in))).astype(np.float32)


def make_weights(seed=1300, profile="sparse", gains=None):
    rng = np.random.default_rng(seed)
    prof = PROFILES[profile]
    gains = gains or prof["gains"]
    w = {}
    for net, layers in spec.NETS.items():
        for l in layers():
            fan_in = l.k ** 3 * l.cin
            g = gains.get(net, {}).get(l.name, 1.0)
            if l.name.endswith("conv1_2") or l.name.endswith("conv2_3"):
                g *= _RES_GAIN
            w["%s/%s/kernel" % (net, l.name)] = _he(rng, spec.kernel_shape(l), fan_in, g)
            if l.bias:
                w["%s/%s/bias" % (net, l.name)] = (rng.standard_normal(l.cout) * 0.05).astype(np.float32)
    # analysis sees a sparse binary cube: lift conv_in so features are O(1)
    w["analysis_transform/conv_in/kernel"] *= np.float32(4.0)
    # positive offset on the scale head keeps Laplace scales away from 0
    w["hyper_decoder/conv4_2/bias"] = (prof["scale_bias"]
                                       + prof["scale_bias_jitter"] * rng.standard_normal(16)).astype(np.float32)
    # EntropyBottleneck(channels=8, init_scale=8, filters=(3,3,3)) — entropy_model.py:41-68
    C, f = 8, (1, 3, 3, 3, 1)
    scale = 8.0 ** (1.0 / 4)
    for i in range(4):
        init = np.log(np.expm1(1.0 / scale / f[i + 1]))
        w["estimator/matrix_%d" % i] = (init + 0.1 * rng.standard_normal((C, f[i + 1], f[i]))).astype(np.float32)
        w["estimator/bais_%d" % i] = rng.uniform(-0.5, 0.5, (C, f[i + 1], 1)).astype(np.float32)
        w["estimator/factor_%d" % i] = (0.2 * rng.standard_normal((C, f[i + 1], 1))).astype(np.float32)
    return w


def make_weights_simple(seed=1300, latent_gain=1.0):
    """Seeded weights of the reference's models/model_simple.py plus its 32-channel EntropyBottleneck
    (train_factorized.py:76-77), keys as in the factorized checkpoints."""
    rng = np.random.default_rng(seed)
    w = {}
    for net, layers in spec.SIMPLE_NETS.items():
        for l in layers():
            # a transposed stride-2 conv feeds each output from k^3 / 8 taps
            w["%s/%s/kernel" % (net, l.name)] = _he(rng, spec.kernel_shape(l), l.k ** 3 * l.cin / (8.0 if l.kind == "tconv" else 1.0))
            if l.bias:
                w["%s/%s/bias" % (net, l.name)] = (rng.standard_normal(l.cout) * 0.05).astype(np.float32)
    w["analysis_transform/conv_1/kernel"] *= np.float32(6.0)          # sparse binary input
    w["analysis_transform/conv_3/kernel"] *= np.float32(latent_gain)
    C, f = 32, (1, 3, 3, 3, 1)
    scale = 8.0 ** (1.0 / 4)
    for i in range(4):
        init = np.log(np.expm1(1.0 / scale / f[i + 1]))
        w["estimator/matrix_%d" % i] = (init + 0.1 * rng.standard_normal((C, f[i + 1], f[i]))).astype(np.float32)
        w["estimator/bais_%d" % i] = rng.uniform(-0.5, 0.5, (C, f[i + 1], 1)).astype(np.float32)
        w["estimator/factor_%d" % i] = (0.2 * rng.standard_normal((C, f[i + 1], 1))).astype(np.float32)
    return w


def make_cloud(seed=1300, res=1024, n_shells=6, rmin=0.05, rmax=0.13, thickness=0.5, oversample=3.0):
    """Union of ellipsoid shells voxelised on a res^3 grid -> unique int32 points [N,3]
    in random (seeded) order, like a scanned cloud.  Defaults give about 0.85 M points /
    about 200 cubes of 64^3 at res=1024 (longdress: 857 966 points, 202 cubes)."""
    rng = np.random.default_rng(seed)
    pts = []
    for _ in range(n_shells):
        c = rng.uniform(0.3, 0.7, 3) * res
        r = rng.uniform(rmin, rmax, 3) * res
        n = int(oversample * 4.0 * np.pi * (r[0] * r[1] + r[1] * r[2] + r[0] * r[2]) / 3)
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        pts.append(c + u * r + rng.uniform(-thickness, thickness, (n, 3)))
    p = np.rint(np.concatenate(pts)).astype(np.int64)
    p = p[np.all((p >= 0) & (p < res), axis=1)]
    key = np.unique((p[:, 0] * res + p[:, 1]) * res + p[:, 2])
    key = key[rng.permutation(len(key))]
    return np.stack([key // (res * res), (key // res) % res, key % res], -1).astype(np.int32)


def make_cubes(seed=1300, n_cubes=8, cube_size=64, occupancy=0.016):
    """Cheap stand-alone cubes (no partition): a wavy sheet per cube, float32
    [B, cs, cs, cs, 1] with roughly `occupancy` ones."""
    rng = np.random.default_rng(seed)
    cs = cube_size
    out = np.zeros((n_cubes, cs, cs, cs, 1), np.float32)
    a, b = np.meshgrid(np.arange(cs), np.arange(cs), indexing="ij")
    for i in range(n_cubes):
        f = rng.uniform(0.05, 0.25, 2)
        ph = rng.uniform(0, 6.28, 2)
        amp = rng.uniform(0.1, 0.3) * cs
        h = cs / 2 + amp * np.sin(f[0] * a + ph[0]) * np.cos(f[1] * b + ph[1])
        h = np.clip(np.rint(h).astype(int), 0, cs - 1)
        axis = i % 3
        idx = [a, b]
        idx.insert(axis, h)
        out[i, idx[0], idx[1], idx[2], 0] = 1.0
        extra = int(max(0.0, occupancy * cs ** 3 - cs * cs))
        if extra:
            q = rng.integers(0, cs, (extra, 3))
            out[i, q[:, 0], q[:, 1], q[:, 2], 0] = 1.0
    return out
