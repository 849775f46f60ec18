"""Offline calibration of pcgcv1_amd/synthetic.py's per-layer gains (run HERE, uses the CPU oracle).

Finds output-layer gains so that, on seeded synthetic cubes, std(y) ~ 2.0,
std(z) ~ 1.5, std(loc) ~ 1.0 and the kernel part of the scale head has
std ~ 0.15.  Prints the dict to paste into synthetic._GAINS.
"""
import numpy as np

from oracle import transform
from pcgcv1_amd import synthetic

TARGETS = {"dense": {"y": 2.0, "z": 1.5, "loc": 1.0, "scale_k": 0.15},
           "sparse": {"y": 0.15, "z": 0.3, "loc": 0.03, "scale_k": 0.02}}


def main(profile):
    TARGET = TARGETS[profile]
    gains = {"analysis_transform": {"conv_out": 1.0}, "synthesis_transform": {"deconv_out": 1.0},
             "hyper_encoder": {"conv3": 1.0}, "hyper_decoder": {"conv4_1": 1.0, "conv4_2": 1.0}}
    x = synthetic.make_cubes(n_cubes=2)
    for it in range(4):
        w = synthetic.make_weights(profile=profile, gains=gains)
        r = transform.rate_terms(w, x)
        bias = w["hyper_decoder/conv4_2/bias"]
        # scale = |conv + bias|; recover the kernel part's std from loc-like statistics of (scale_raw - bias)
        sk = np.std(r["scale"] - bias)  # approximate (abs folds a small tail only)
        print(it, {k: float(np.std(r[k])) for k in ("y", "z", "loc")}, "scale_k", float(sk),
              "scale min/mean/max", float(r["scale"].min()), float(r["scale"].mean()), float(r["scale"].max()),
              "bpp", r["bpp_y"], r["bpp_z"])
        gains["analysis_transform"]["conv_out"] *= TARGET["y"] / np.std(r["y"])
        gains["hyper_encoder"]["conv3"] *= TARGET["z"] / np.std(r["z"])
        gains["hyper_decoder"]["conv4_1"] *= TARGET["loc"] / np.std(r["loc"])
        gains["hyper_decoder"]["conv4_2"] *= TARGET["scale_k"] / sk
    print({n: {k: round(float(v), 4) for k, v in g.items()} for n, g in gains.items()})


if __name__ == "__main__":
    import sys
    main(sys.argv[1])
