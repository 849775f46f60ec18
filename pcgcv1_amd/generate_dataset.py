"""Training-set generator with the reference's interface (generate_dataset.py:11-38): every .ply under INPUT_DIR is
partitioned into cube_size^3 cubes with at least 20 points (dataprocess.inout_points.load_points, the same partition
the codec uses), the cubes are shuffled and each one is written as a uint8 [n,3] array of in-cube coordinates named
<ply stem>_<i>n.<ext>.

The reference stores each cube as the HDF5 dataset 'data' (generate_dataset.py:27-29).  h5py is not part of this
image: the default container here is .npy with the same dtype, shape and content; `fmt="h5"` writes HDF5 through h5py
when it is importable and through dataprocess/h5min.py (the same file structures, pure Python) otherwise.
pcgcv1_amd.train_hyper reads both (load_cube_points).
"""
import glob
import os
import random

import numpy as np

from .dataprocess.inout_points import load_points


def write_cube(path_stem, points, fmt="npy"):
    points = np.ascontiguousarray(points).astype("uint8")
    if fmt == "h5":
        try:
            import h5py                               # generate_dataset.py:27-29
        except ImportError:                           # not in this image: the minimal writer lays out the same structures
            from .dataprocess import h5min
            h5min.write_dataset(path_stem + ".h5", points, "data")
            return path_stem + ".h5"
        with h5py.File(path_stem + ".h5", "w") as h:
            h.create_dataset("data", data=points, shape=points.shape)
        return path_stem + ".h5"
    np.save(path_stem + ".npy", points)
    return path_stem + ".npy"


def generate_dataset(INPUT_DIR, OUTPUT_DIR, DATA_NUM, cube_size=64, fmt="npy", seed=None):
    """-> list of the files written.  Stops after the first ply that takes the count past DATA_NUM (generate_dataset.py:35-36)."""
    if cube_size > 256:
        raise ValueError("in-cube coordinates are stored as uint8 (generate_dataset.py:23): cube_size must be <= 256")
    rnd = random.Random(seed)
    plydirs = sorted(glob.glob(os.path.join(INPUT_DIR, "*.ply")))
    rnd.shuffle(plydirs)
    os.makedirs(OUTPUT_DIR, exist_ok=True)
    written = []
    for filename in plydirs:
        set_points, _ = load_points(filename, cube_size=cube_size, min_num=20)
        set_points = list(set_points)
        rnd.shuffle(set_points)
        stem = os.path.splitext(os.path.basename(filename))[0]
        for i, points in enumerate(set_points):
            written.append(write_cube(os.path.join(OUTPUT_DIR, "%s_%dn" % (stem, i)), points, fmt))
        if len(written) > DATA_NUM:
            break
    return written


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input_dir")
    ap.add_argument("output_dir")
    ap.add_argument("--data_num", type=float, default=1e6)
    ap.add_argument("--cube_size", type=int, default=64)
    ap.add_argument("--fmt", choices=("npy", "h5"), default="npy")
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args(argv)
    files = generate_dataset(a.input_dir, a.output_dir, a.data_num, a.cube_size, a.fmt, a.seed)
    print("cubes written:", len(files))


if __name__ == "__main__":
    main()
