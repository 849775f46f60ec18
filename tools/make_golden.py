"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's numpy-only modules.

Run in the build container only (needs /root/reference; the GPU box never has
it):      python tools/make_golden.py

Imports /root/reference/{process.py, dataprocess/inout_points.py,
dataprocess/inout_bitstream.py, myutils/pc_error_wrapper.py} (pure numpy,
importable without TensorFlow) and the prebuilt myutils/tmc3, myutils/pc_error_d
tools, feeds them seeded synthetic inputs and stores inputs + outputs.  The
fixtures are data (arrays / bytes), never reference source text.
"""
import io
import contextlib
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_cloud(seed, res, n):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.35, 0.65, 3) * res
    r = rng.uniform(0.15, 0.3, 3) * res
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    p = np.rint(c + u * r).astype(np.int64)
    p = p[np.all((p >= 0) & (p < res), axis=1)]
    _, first = np.unique(p, axis=0, return_index=True)
    return p[np.sort(first)].astype(np.int32)          # unique, original (random) order


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir(REF)                                       # gpcc_wrapper.py:11 uses the relative path myutils/tmc3
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from dataprocess import inout_points as rp
    from dataprocess import inout_bitstream as rb
    import process as rproc
    tmp = tempfile.mkdtemp(prefix="golden_")
    quiet = contextlib.redirect_stdout(io.StringIO())

    # ---------------------------------------------------------------- partition / preprocess
    cases = {}
    for name, seed, res, n, cube, min_num, scale in [
        ("a", 11, 256, 6000, 64, 20, 1.0),
        ("b", 12, 128, 3000, 32, 64, 1.0),
        ("c", 13, 256, 9000, 64, 64, 0.5),
        ("d", 14, 512, 20000, 64, 30, 1.0),
    ]:
        pts = seeded_cloud(seed, res, n)
        ply = os.path.join(tmp, "in_%s.ply" % name)
        rp.write_ply_data(ply, pts)
        with quiet:
            cubes, cube_positions, points_numbers = rproc.preprocess(ply, scale, cube, min_num)
        if scale == 1.0:
            set_points, cube_positions2 = rp.load_points(ply, cube, min_num)
            assert np.array_equal(cube_positions, cube_positions2)
        occ = [np.flatnonzero(c).astype(np.int32) for c in cubes]
        cases.update({
            name + "_points": pts, name + "_args": np.array([cube, min_num, scale]),
            name + "_ply": np.frombuffer(open(ply, "rb").read(), np.uint8),
            name + "_cube_positions": np.asarray(cube_positions, np.int64),
            name + "_points_numbers": points_numbers,
            name + "_occ_flat": np.concatenate(occ), name + "_occ_lens": np.array([len(o) for o in occ]),
        })
        # identity round trip through postprocess (threshold on the {0,1} cubes)
        out_ply = os.path.join(tmp, "rec_%s.ply" % name)
        with quiet:
            rproc.postprocess(out_ply, cubes, points_numbers, cube_positions, scale, cube, 1.0)
        cases[name + "_rec_ply"] = np.frombuffer(open(out_ply, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "partition.npz"), **cases)

    # ---------------------------------------------------------------- select_voxels / voxels2points / save_points
    rng = np.random.default_rng(21)
    sel = {}
    vols = (rng.standard_normal((5, 16, 16, 16, 1)) * 3).astype(np.float32)
    vols[1] = np.round(vols[1])                          # heavy ties
    vols[2] = -5.0 - np.abs(vols[2])                     # nothing above init_thres=-2 -> falls back to all voxels
    vols[3, :2] = 7.25                                   # plateau of equal maxima
    nums = np.array([100, 40, 17, 300, 0], np.uint16)    # includes k = 0
    for rho in (1.0, 1.1, 0.5):
        sel["mask_rho%g" % rho] = rp.select_voxels(vols, nums, rho).astype(np.uint8)
    sel["mask_fixed0"] = rp.select_voxels(vols, nums, 1.0, fixed_thres=0.0).astype(np.uint8)
    sel["vols"], sel["nums"] = vols, nums
    pts = rp.voxels2points(sel["mask_rho1"])
    sel["v2p_flat"] = np.concatenate(pts).astype(np.int32)
    sel["v2p_lens"] = np.array([len(p) for p in pts])
    pos = np.array([[3, 0, 1], [0, 2, 2], [1, 1, 0], [2, 2, 2], [0, 0, 1]])
    ply = os.path.join(tmp, "merge.ply")
    rp.save_points(pts, pos, ply, 16)
    sel["merge_positions"] = pos
    sel["merge_ply"] = np.frombuffer(open(ply, "rb").read(), np.uint8)
    # float writer (scale != 1 path of postprocess, process.py:76-78)
    fpts = (np.array([[0, 1, 2], [3, 4, 5], [100, 7, 1023]], np.int32).astype("float32") * float(1 / 0.375))
    ply = os.path.join(tmp, "float.ply")
    rp.write_ply_data(ply, fpts)
    sel["float_points"] = fpts
    sel["float_ply"] = np.frombuffer(open(ply, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "select.npz"), **sel)

    # ---------------------------------------------------------------- container format
    rng = np.random.default_rng(31)
    B = 6
    lens = [3, 255, 256, 0 + 40, 1000, 17]
    y_strings = [bytes(rng.integers(1, 256, l, dtype=np.uint8)) for l in lens]
    y_min_vs = np.array([-3, -15, 0, -1, -7, -2], np.int32)
    y_max_vs = np.array([2, 15, 1, 0, 9, 3], np.int32)
    y_shape = np.array([1, 16, 16, 16, 16], np.int32)
    z_string = bytes(rng.integers(0, 256, 77, dtype=np.uint8))
    z_shape = np.array([B, 8, 8, 8, 8], np.int32)
    points_numbers = np.array([78, 4246, 11450, 64, 65535, 300], np.uint16)
    cube_positions = np.array([[1, 2, 3], [0, 0, 0], [15, 3, 7], [4, 4, 4], [2, 9, 1], [7, 7, 0]])
    root = os.path.join(tmp, "bits")
    with quiet:
        sizes = rb.write_binary_files_hyper("g", y_strings, z_string, points_numbers, cube_positions,
                                            y_min_vs, y_max_vs, y_shape, -6, 5, z_shape, rootdir=root)
    bits = {"sizes": np.array(sizes)}
    for ext in ("strings", "strings_head", "strings_hyper", "pointnums", "cubepos"):
        bits[ext] = np.frombuffer(open(os.path.join(root, "g." + ext), "rb").read(), np.uint8)
    bits.update(y_lens=np.array(lens), y_concat=np.frombuffer(b"".join(y_strings), np.uint8),
                y_min_vs=y_min_vs, y_max_vs=y_max_vs, y_shape=y_shape, z_string=np.frombuffer(z_string, np.uint8),
                z_min_v=np.array(-6), z_max_v=np.array(5), z_shape=z_shape, points_numbers=points_numbers,
                cube_positions=cube_positions)
    # the reference reader cannot parse a head that mixes <=255 and >255 lengths on numpy 2.x
    # (inout_bitstream.py:168-174), so decode only the tmc3 part to record the position order it yields.
    from myutils.gpcc_wrapper import gpcc_decode
    gpcc_decode(os.path.join(root, "g.cubepos"), os.path.join(root, "g_dec.ply"))
    bits["cubepos_decoded"] = rp.load_ply_data(os.path.join(root, "g_dec.ply"))
    # a head the reference reader CAN parse (all lengths <= 255): record its outputs
    lens2 = [3, 255, 40, 17]
    ys2 = [bytes(rng.integers(1, 256, l, dtype=np.uint8)) for l in lens2]
    with quiet:
        rb.write_binary_files_hyper("h", ys2, z_string, points_numbers[:4], cube_positions[:4],
                                    y_min_vs[:4], y_max_vs[:4], y_shape, -6, 5, np.array([4, 8, 8, 8, 8]),
                                    rootdir=root)
        r = rb.read_binary_files_hyper("h", rootdir=root)
    bits["h_strings_head"] = np.frombuffer(open(os.path.join(root, "h.strings_head"), "rb").read(), np.uint8)
    bits["h_strings"] = np.frombuffer(open(os.path.join(root, "h.strings"), "rb").read(), np.uint8)
    bits["h_lens"] = np.array(lens2)
    bits["h_read_y_min_vs"], bits["h_read_y_max_vs"] = r[4], r[5]
    bits["h_read_y_shape"], bits["h_read_z"] = r[6], np.array([r[7], r[8]])
    bits["h_read_cube_positions"] = r[3]
    np.savez_compressed(os.path.join(OUT, "bitstream_hyper.npz"), **bits)

    # ---------------------------------------------------------------- pc_error_d (D1) known answers
    d1 = {}
    for i, (seed, res, n, drop, jit) in enumerate([(41, 256, 5000, 0.1, 1), (42, 1024, 20000, 0.02, 2)]):
        a = seeded_cloud(seed, res, n)
        rng = np.random.default_rng(seed + 100)
        b = a[rng.random(len(a)) > drop].copy()
        b += rng.integers(-jit, jit + 1, b.shape).astype(np.int32) * (rng.random(b.shape) < 0.2)
        b = np.unique(np.clip(b, 0, res - 1), axis=0).astype(np.int32)
        fa, fb = os.path.join(tmp, "a%d.ply" % i), os.path.join(tmp, "b%d.ply" % i)
        rp.write_ply_data(fa, a)
        rp.write_ply_data(fb, b)
        out = subprocess.run(["myutils/pc_error_d", "-a", fa, "-b", fb, "--hausdorff=1", "-r", str(res - 1)],
                             capture_output=True, text=True).stdout
        vals = {}
        for line in out.splitlines():
            for key in ("mse1      (p2point)", "mse2      (p2point)", "mseF      (p2point)",
                        "mse1,PSNR (p2point)", "mse2,PSNR (p2point)", "mseF,PSNR (p2point)",
                        "h.       1(p2point)", "h.       2(p2point)", "h.        (p2point)"):
                if line.strip().startswith(key):
                    vals[key] = float(line.split(":")[-1])
        d1["a%d" % i], d1["b%d" % i] = a, b
        d1["res%d" % i] = np.array(res)
        d1["keys%d" % i] = np.array(sorted(vals))
        d1["vals%d" % i] = np.array([vals[k] for k in sorted(vals)])
        print("pc_error", i, vals)
    np.savez_compressed(os.path.join(OUT, "pc_error_d1.npz"), **d1)

    # ---------------------------------------------------------------- pc_error_d with normals (D2, point-to-plane)
    # the call of myutils/pc_error_wrapper.py:46-51 (-n = the original cloud's normals); clouds with many
    # equal-distance neighbours (voxel grids) plus hand-made tie cases
    def write_ply_normals(fn, pts, normals=None):
        with open(fn, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(pts))
            if normals is not None:
                f.write("property float nx\nproperty float ny\nproperty float nz\n")
            f.write("end_header\n")
            for i in range(len(pts)):
                row = "%d %d %d" % tuple(pts[i])
                if normals is not None:
                    row += " %.9g %.9g %.9g" % tuple(normals[i])
                f.write(row + "\n")

    d2 = {}
    cases = []
    for seed, res, n, drop, jit in [(51, 128, 4000, 0.15, 1), (52, 512, 30000, 0.05, 2)]:
        a = seeded_cloud(seed, res, n)
        rng = np.random.default_rng(seed + 100)
        c = a.mean(0)
        na = (a - c) / np.linalg.norm(a - c, axis=1, keepdims=True)
        na = (na + 0.05 * rng.standard_normal(na.shape)).astype(np.float32)           # not unit length on purpose
        b = a[rng.random(len(a)) > drop].copy()
        b += rng.integers(-jit, jit + 1, b.shape).astype(np.int32) * (rng.random(b.shape) < 0.3)
        b = np.unique(np.clip(b, 0, res - 1), axis=0).astype(np.int32)
        cases.append((a, na, b, res))
    cases.append((np.array([[0, 0, 0], [2, 0, 0], [0, 0, 9]], np.int32), np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32),
                  np.array([[1, 0, 0], [0, 0, 8], [0, 0, 1], [2, 0, 1]], np.int32), 16))
    cases.append((np.array([[0, 0, 0], [0, 0, 3]], np.int32), np.array([[1, 0, 0], [0, 0, 1]], np.float32),
                  np.array([[1, 0, 0], [0, 0, 1]], np.int32), 16))
    keys = ["mse1      (p2point)", "mse2      (p2point)", "mseF      (p2point)", "mseF,PSNR (p2point)",
            "mse1      (p2plane)", "mse2      (p2plane)", "mseF      (p2plane)", "mseF,PSNR (p2plane)",
            "h.       1(p2plane)", "h.       2(p2plane)", "h.        (p2plane)"]
    for i, (a, na, b, res) in enumerate(cases):
        fa, fb = os.path.join(tmp, "na%d.ply" % i), os.path.join(tmp, "nb%d.ply" % i)
        write_ply_normals(fa, a, na)
        write_ply_normals(fb, b)
        out = subprocess.run(["myutils/pc_error_d", "-a", fa, "-b", fb, "-n", fa, "--hausdorff=1", "--resolution=%d" % (res - 1)],
                             capture_output=True, text=True).stdout
        vals = {}
        for line in out.splitlines():
            for key in keys:
                if line.strip().startswith(key):
                    vals[key] = float(line.split(":")[-1])
        assert sorted(vals) == sorted(keys), out
        d2["a%d" % i], d2["na%d" % i], d2["b%d" % i], d2["res%d" % i] = a, na, b, np.array(res)
        d2["vals%d" % i] = np.array([vals[k] for k in keys])
        print("pc_error D2", i, vals)
    d2["keys"] = np.array(keys)
    d2["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "pc_error_d2.npz"), **d2)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
