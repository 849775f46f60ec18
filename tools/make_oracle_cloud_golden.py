"""Full BASELINE configs[1] cloud through the CPU side only, once, in the build container -> tests/golden/oracle_<rate>_cloud1300.npz

    python tools/make_oracle_cloud_golden.py [--rate a6.00b3.00] [--seed 1300] [--scale 0.625] [--normals]

--scale: the rate section's down-scale (eval_ablation_studies.py:71: R1 = a0.75b3 at 5/8), through the reference's own
process.preprocess / postprocess.  --normals: the input ply carries the radial normals the config-3 test writes
(tests/test_gpu_parity.py::test_config3_*), pc_error_d gets `-n` like myutils/pc_error_wrapper.py:48-53 passes it, and the
fixture holds the point-to-plane (D2) numbers too.

What runs (nothing of the HIP path):
  * the held-out cloud of bench.py (synthetic.make_cloud(seed 1300): 828 225 points) written as an ASCII ply;
  * the REFERENCE's own numpy modules, imported from /root/reference (they do not need TensorFlow):
    process.preprocess (partition + voxelisation, process.py:16-52), process.postprocess (top-k + merge + ply,
    process.py:54-82), dataprocess.inout_bitstream.write_binary_files_hyper (container incl. the prebuilt tmc3,
    inout_bitstream.py:75-141) and the prebuilt myutils/pc_error_d (D1);
  * between them, where the reference needs TensorFlow 1.13: the CPU restatement oracle/transform.py
    (compress_hyper / decompress_hyper: torch-CPU conv3d one cube per call, numpy entropy models, oracle/coder.c) with the
    a6b3 checkpoint committed under checkpoints/hyper/.

The fixture is data: per-cube string lengths / ranges / strings, the z string, the rounded latents, per-cube
reconstruction thresholds and point-set checksums, container file sizes, bpp, pc_error_d's D1 numbers.  The `-m gpu` test
tests/test_trained_checkpoints.py::test_full_cloud_hip_vs_oracle_golden compares the HIP path with it on the GPU box
(where neither /root/reference nor the minutes of CPU time are available).
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rate", default="a6.00b3.00")
    ap.add_argument("--seed", type=int, default=1300)
    ap.add_argument("--limit", type=int, default=0, help="first N cubes only (debugging)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--normals", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    from oracle import points as opoints
    from oracle import transform as otransform
    from pcgcv1_amd import checkpoint, synthetic
    w = checkpoint.load(os.path.join(ROOT, "checkpoints", "hyper", a.rate))
    pts = synthetic.make_cloud(seed=a.seed)
    os.chdir(REF)                                       # gpcc_wrapper.py:11 uses the relative path myutils/tmc3
    sys.path.insert(0, REF)
    from dataprocess import inout_points as rp
    import process as rproc
    tmp = tempfile.mkdtemp(prefix="oracle_cloud_")
    quiet = contextlib.redirect_stdout(io.StringIO())
    ply = os.path.join(tmp, "cloud_vox10_%d.ply" % a.seed)
    if a.normals:                                       # the same text the config-3 test writes
        c = pts.mean(0)
        nrm = (pts - c) / np.maximum(np.linalg.norm(pts - c, axis=1, keepdims=True), 1e-9)
        with open(ply, "w") as fh:
            fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                     "property float nx\nproperty float ny\nproperty float nz\nend_header\n" % len(pts))
            np.savetxt(fh, np.concatenate([pts.astype(np.float64), nrm], 1), fmt="%d %d %d %.6f %.6f %.6f")
    else:
        rp.write_ply_data(ply, pts)
    t0 = time.time()
    with quiet:
        cubes, cube_positions, points_numbers = rproc.preprocess(ply, a.scale, 64, 64)
    print("reference preprocess: %d cubes in %.1f s" % (len(cubes), time.time() - t0), flush=True)
    if a.limit:
        cubes, points_numbers = cubes[:a.limit], points_numbers[:a.limit]      # debugging: no merge / D1 below
    B = len(cubes)
    tm = {}
    t0 = time.time()
    out = otransform.compress_hyper(cubes.astype(np.float32), w, timers=tm)
    print("oracle compress_hyper: %.1f s %s" % (time.time() - t0, {k: round(v, 2) for k, v in tm.items()}), flush=True)
    y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape = out
    tm = {}
    t0 = time.time()
    x_tilde = otransform.decompress_hyper(*out, w, timers=tm)
    print("oracle decompress_hyper: %.1f s %s" % (time.time() - t0, {k: round(v, 2) for k, v in tm.items()}), flush=True)
    # rounded latents (what the strings carry), from the strings themselves
    from oracle import entropy as oent
    from oracle import nets as onets
    eb = onets.sub(w, "estimator")
    z_hat = oent.eb_decompress(eb, z_string, z_min_v, z_max_v, z_shape)
    whd = onets.sub(w, "hyper_decoder")
    y_hat = np.empty((B, 16, 16, 16, 16), np.int8)
    for i in range(B):
        loc, scale = onets.hyper_decoder(whd, z_hat[i:i + 1])
        scale = np.maximum(scale, otransform.LOWER_BOUND)
        y_hat[i] = oent.sc_decompress(y_strings[i], loc, scale, y_min_vs[i], y_max_vs[i], y_shape)[0]
    sizes = {}
    # the four numpy-packed files through the pinned restatement (byte-exact vs the reference writer in
    # tests/golden/bitstream_hyper.npz); .cubepos through the prebuilt tmc3 exactly as inout_bitstream.py:117-120 calls it
    from oracle import bitstream as obit
    sizes["strings"] = len(obit.pack_strings(y_strings))
    sizes["strings_head"] = len(obit.pack_strings_head(y_strings, y_min_vs, y_max_vs, y_shape))
    sizes["strings_hyper"] = len(obit.pack_strings_hyper(z_string, z_min_v, z_max_v, z_shape))
    sizes["pointnums"] = len(obit.pack_pointnums(points_numbers))
    try:
        from myutils.gpcc_wrapper import gpcc_encode
        cp_ply = os.path.join(tmp, "cubepos.ply")
        rp.write_ply_data(cp_ply, np.asarray(cube_positions).astype("uint8"))
        with quiet:
            gpcc_encode(cp_ply, os.path.join(tmp, "cloud.cubepos"))
        sizes["cubepos_tmc3"] = os.path.getsize(os.path.join(tmp, "cloud.cubepos"))
    except Exception as e:                              # noqa: BLE001
        print("tmc3 failed:", e)
        sizes["cubepos_tmc3"] = -1
    # the reference's postprocess + pc_error_d
    rec_ply = os.path.join(tmp, "cloud_rec.ply")
    t0 = time.time()
    if a.limit:
        rp.write_ply_data(rec_ply, pts)
    else:
        with quiet:
            rproc.postprocess(rec_ply, x_tilde, points_numbers, cube_positions, a.scale, 64, 1.0)
    rec = rp.load_ply_data(rec_ply)
    print("reference postprocess: %d points in %.1f s" % (len(rec), time.time() - t0), flush=True)
    t0 = time.time()
    txt = subprocess.run(["myutils/pc_error_d", "-a", ply, "-b", rec_ply] + (["-n", ply] if a.normals else []) +
                         ["--hausdorff=1", "-r", "1023"], capture_output=True, text=True).stdout
    vals = {}
    keys = ("mse1      (p2point)", "mse2      (p2point)", "mseF      (p2point)",
            "mse1,PSNR (p2point)", "mse2,PSNR (p2point)", "mseF,PSNR (p2point)")
    if a.normals:
        keys += ("mse1      (p2plane)", "mse2      (p2plane)", "mseF      (p2plane)",
                 "mse1,PSNR (p2plane)", "mse2,PSNR (p2plane)", "mseF,PSNR (p2plane)")
    for line in txt.splitlines():
        for key in keys:
            if line.strip().startswith(key):
                vals[key] = float(line.split(":")[-1])
    print("pc_error_d: %.1f s %s" % (time.time() - t0, vals), flush=True)
    # per-cube reconstruction: threshold, point count and a checksum of the cube's local point list
    masks = opoints.select_voxels(x_tilde, points_numbers, 1.0)
    thr = np.array([opoints.adaptive_threshold(x_tilde[i], int(points_numbers[i])) for i in range(B)], np.float32)
    rec_counts = np.array([int(m.sum()) for m in masks], np.int32)
    rec_crc = np.array([zlib.crc32(np.flatnonzero(m.reshape(-1)).astype(np.int32).tobytes()) for m in masks], np.uint32)
    npts = float(len(pts))
    nbytes_latents = sizes["strings"] + sizes["strings_hyper"] - 12
    gold = dict(
        seed=np.array(a.seed), rate=np.array(a.rate), scale=np.array(a.scale), n_points=np.array(len(pts)), n_cubes=np.array(B),
        cube_positions=np.asarray(cube_positions, np.int32), points_numbers=np.asarray(points_numbers, np.uint16),
        y_lens=np.array([len(s) for s in y_strings], np.int32), y_min_vs=np.asarray(y_min_vs, np.int32), y_max_vs=np.asarray(y_max_vs, np.int32),
        y_blob=np.frombuffer(b"".join(bytes(s) for s in y_strings), np.uint8), y_shape=np.asarray(y_shape, np.int32),
        z_string=np.frombuffer(bytes(z_string), np.uint8), z_min_v=np.array(z_min_v), z_max_v=np.array(z_max_v), z_shape=np.asarray(z_shape, np.int32),
        y_hat=y_hat, z_hat=np.rint(z_hat).astype(np.int8),
        thresholds=thr, rec_counts=rec_counts, rec_crc=rec_crc, n_points_out=np.array(len(rec)),
        x_tilde_absmax=np.abs(x_tilde).reshape(B, -1).max(1).astype(np.float32),
        x_tilde_sum=x_tilde.reshape(B, -1).sum(1, dtype=np.float64),
        file_keys=np.array(sorted(sizes)), file_sizes=np.array([sizes[k] for k in sorted(sizes)], np.int64),
        bpp_latents=np.array(8.0 * (sum(len(s) for s in y_strings) + len(z_string)) / npts),
        bpp_4files=np.array(8.0 * (sizes["strings"] + sizes["strings_head"] + sizes["strings_hyper"] + sizes["pointnums"]) / npts),
        d1_keys=np.array(sorted(vals)), d1_vals=np.array([vals[k] for k in sorted(vals)]),
    )
    assert nbytes_latents == sum(len(s) for s in y_strings) + len(z_string)
    name = "oracle_%s_cloud%d%s%s.npz" % (a.rate.replace(".00", ""), a.seed, "_s%g" % a.scale if a.scale != 1.0 else "",
                                          "_first%d" % a.limit if a.limit else "")
    dst = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(dst, **gold)
    print("wrote %s (%.0f KB): bpp_latents %.5f bpp_4files %.5f D1 %s" % (dst, os.path.getsize(dst) / 1e3, gold["bpp_latents"],
                                                                     gold["bpp_4files"], vals.get("mseF,PSNR (p2point)")))


if __name__ == "__main__":
    main()
