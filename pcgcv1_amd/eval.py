"""One rate point of the reference's RD harness (eval.py:77-113 `test_hyper` + 194-207): compress, write the
container, read it back, decompress, classify with the adaptive threshold, and report bpp (file bytes over input
points, itemised like eval.py:102-111) and D1 PSNR (pcgcv1_amd.metrics, pinned to pc_error_d).  Unlike the
reference no "cheat" substitution of the encoder-side reconstruction is needed (eval.py:96-100): the decoder is
bit-reproducible.
"""
import os
import tempfile

import numpy as np

from . import metrics
from .dataprocess import inout_bitstream as bs
from .dataprocess import inout_points as iop
from .process import postprocess_points, preprocess_points
from .transform import compress_hyper, decompress_hyper


def test_hyper(points, model, ckpt_dir, scale=1.0, cube_size=64, min_num=64, rho=1.0, resolution=1023, rootdir=None):
    points = np.asarray(points)
    cubes, cube_positions, points_numbers = preprocess_points(points, scale, cube_size, min_num)
    stream = compress_hyper(cubes, model, ckpt_dir)
    y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape = stream
    own_tmp = rootdir is None
    rootdir = rootdir or tempfile.mkdtemp(prefix="pcgc_eval_")
    sizes = bs.write_binary_files_hyper("x", y_strings, z_string, points_numbers, cube_positions, y_min_vs, y_max_vs, y_shape,
                                        z_min_v, z_max_v, z_shape, rootdir=rootdir, verbose=False)
    r = bs.read_binary_files_hyper("x", rootdir=rootdir)
    cubes_d = decompress_hyper(r[0], r[4], r[5], r[6], r[1], r[7], r[8], r[9], model, ckpt_dir)
    rec = postprocess_points(cubes_d, r[2], r[3], scale, cube_size, rho)
    n = float(len(points))
    names = ("strings", "strings_head", "strings_hyper", "pointnums", "cubepos")
    out = {"bpp": metrics.bpp(sum(sizes), n), "n_cubes": int(cubes.shape[0]), "n_points_in": int(n), "n_points_out": int(len(rec))}
    out.update({"bpp_" + k: metrics.bpp(v, n) for k, v in zip(names, sizes)})
    out["d1_psnr"] = metrics.d1_psnr(points.astype(np.int32), np.rint(rec).astype(np.int32), resolution)
    if own_tmp:
        for k in names:
            os.remove(os.path.join(rootdir, "x." + k))
        os.rmdir(rootdir)
    return out
