"""The reference's RD harness (eval.py): `test_hyper` / `test_factorized` = one rate point (eval.py:77-113 / 45-75: compress,
write the container, read it back, decompress, bpp itemised like 102-111 / 64-71), `eval` = the loop over the rate sections of a
config .ini with the three reconstructions rho = 1 / rho_d1 / rho_d2 and the pc_error table per rate (160-215),
written as <rootdir>/<name>.csv with the reference's column names.  D1 / D2 come from pcgcv1_amd.metrics (device
kernels pinned to the prebuilt pc_error_d).  A rate section without rho_d1 / rho_d2 gets them from the reference's
search (eval_ablation_studies.py:152-205: `select_optimal_rho` over its two ladders, written back into the .ini);
`set_default_config` writes the reference's default .ini (eval_ablation_studies.py:45-80, hyper: R1 … R7).  Unlike the reference no "cheat" substitution of the encoder-side
reconstruction is needed (eval.py:96-100): the decoder is bit-reproducible.  The matplotlib plot (136-157) is not
reproduced; the csv holds every plotted series.
"""
import configparser
import csv
import importlib
import os
import tempfile

import numpy as np

from . import metrics
from .dataprocess import inout_bitstream as bs
from .dataprocess import inout_points as iop
from .process import postprocess_points, preprocess_points
from .transform import compress_factorized, compress_hyper, compress_hyper_ahead, decompress_factorized, decompress_hyper


def _d1_of(points, rec, resolution):
    """mseF PSNR (p2point) of a reconstruction: on the device for a cloud on the integer grid, metrics.pc_error_off_grid for
    one scaled back by 1 / scale (fractional coordinates, which pc_error measures as they are)."""
    if metrics._off_grid(rec):
        return metrics.pc_error(points, rec, None, resolution)["mseF,PSNR (p2point)"]
    return metrics.d1_psnr(points.astype(np.int32), np.rint(rec).astype(np.int32), resolution)


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
    out["d1_psnr"] = _d1_of(points, rec, resolution)
    if own_tmp:
        for k in names:
            os.remove(os.path.join(rootdir, "x." + k))
        os.rmdir(rootdir)
    return out


def test_factorized(points, model, ckpt_dir, scale=1.0, cube_size=64, min_num=64, rho=1.0, resolution=1023, rootdir=None):
    """eval.py:45-75: one rate point of --mode=factorized (analysis -> factorized prior on y -> three-file container ->
    synthesis); the bpp itemisation has no hyper / head terms (eval.py:69-70)."""
    points = np.asarray(points)
    cubes_d, cube_positions, points_numbers, n, bpps = rate_point(points, model, ckpt_dir, scale, cube_size, min_num, rootdir=rootdir,
                                                                 mode="factorized")
    rec = postprocess_points(cubes_d, points_numbers, cube_positions, scale, cube_size, rho)
    out = {"bpp": bpps[0], "bpp_strings": bpps[1], "bpp_strings_hyper": bpps[2], "bpp_strings_head": bpps[3], "bpp_pointnums": bpps[4],
           "bpp_cubepos": bpps[5], "n_cubes": int(len(points_numbers)), "n_points_in": int(n), "n_points_out": int(len(rec))}
    out["d1_psnr"] = _d1_of(points, rec, resolution)
    return out


RHOS_D1 = [0.8, 0.9, 1.0, 1.02, 1.05, 1.10, 1.15, 1.2, 1.25, 1.30, 1.40, 1.50, 1.75, 2.0, 2.5, 3.0]     # eval_ablation_studies.py:184
RHOS_D2 = [1.0, 0.98, 0.95, 0.92, 0.90, 0.88, 0.85, 0.82, 0.80, 0.75, 0.70, 0.65, 0.50, 0.40, 0.30]          # :196
HYPER_RATES = [("R1", 5 / 8., 0.75), ("R2", 1.0, 0.75), ("R3", 1.0, 2.0), ("R4", 1.0, 3.5), ("R5", 1.0, 6.0), ("R6", 1.0, 10.0),
               ("R7", 1.0, 16.0)]                                                                           # :71-77
FACTORIZED_RATES = [("R1", 0.625, 2.0), ("R2", 1.0, 2.0), ("R3", 1.0, 4.0), ("R4", 1.0, 6.0), ("R5", 1.0, 10.0), ("R6", 1.0, 16.0)]  # :54-60
FACTORIZED_RATES_SIMPLE = [("R%d" % k, 1.0, float(k)) for k in range(1, 7)]                                # :61-67 (simple/a<k>b3)


def select_optimal_rho(item, rhos, measure, log=None):
    """eval_ablation_studies.py:152-172, statement for statement: walk the ladder, stop at the first step whose PSNR is
    below the running maximum, return the last rho before it.  As in the reference the running maximum starts at 0 and
    never includes the FIRST ladder entry, so the second entry always replaces the first (sic).
    `measure(rho)` -> the pc_error dict of the reconstruction at that rho (postprocess + pc_error in the reference)."""
    optimal_rho, max_psnr = None, 0.0
    for i, rho in enumerate(rhos):
        psnr = float(measure(rho)[item])
        if log is not None:
            log.append((item, i, rho, psnr))
        if i == 0:
            max_psnr = 0.0
            optimal_rho = rho
        else:
            max_psnr = max(psnr, max_psnr)
        if psnr < max_psnr:
            break
        optimal_rho = rho
    return optimal_rho


def cfg_post_process(config, config_file, rate, measure, have_normals=True, log=None):
    """eval_ablation_studies.py:175-205: rho_d1 / rho_d2 of a rate section — read from the .ini when present, else searched
    with select_optimal_rho and WRITTEN BACK to the file.  Without normals on the input there is no point-to-plane
    metric to search on (the reference's pc_error call would fail): rho_d2 = 1.0 then, and it is not written."""
    if config.has_option(rate, "rho_d1"):
        rho_d1 = float(config.get(rate, "rho_d1"))
    else:
        rho_d1 = select_optimal_rho("mseF,PSNR (p2point)", RHOS_D1, measure, log)
        config.set(rate, "rho_d1", str(rho_d1))
        with open(config_file, "w") as f:
            config.write(f)
    if config.has_option(rate, "rho_d2"):
        rho_d2 = float(config.get(rate, "rho_d2"))
    elif not have_normals:
        rho_d2 = 1.0
    else:
        rho_d2 = select_optimal_rho("mseF,PSNR (p2plane)", RHOS_D2, measure, log)
        config.set(rate, "rho_d2", str(rho_d2))
        with open(config_file, "w") as f:
            config.write(f)
    return rho_d1, rho_d2


def set_default_config(input_file, cfg_rootdir, resolution, mode="hyper", cube_size=64, ckpt_root=None,
                       modelname="models.model_voxception"):
    """eval_ablation_studies.py:45-80: <cfg_rootdir>/<name>.ini with DEFAULT {cube_size, min_num, resolution} and the rate
    sections — hyper: R1 (a0.75b3 at scale 5/8) … R7 (a16b3); factorized: R1 (a2b3 at 0.625) … R6 (a16b3) for
    model_voxception, simple/a1b3 … simple/a6b3 at scale 1 for model_simple.  A checkpoint directory is looked up under the
    reference's name (a6b3) first, then under the name its train scripts give it (a6.00b3.00, train_hyper.py:272).  An
    existing file is read, not overwritten.  (The reference writes the file for --mode=hyper only — its `config.write`
    sits inside that branch, :78 — and re-derives the factorized defaults on every run; here both modes are written.)"""
    if mode not in ("hyper", "factorized"):
        raise ValueError("set_default_config: mode must be 'hyper' or 'factorized' (got %r)" % (mode,))
    if ckpt_root is None:
        ckpt_root = "./checkpoints/" + mode
    filename = os.path.split(input_file)[-1][:-4]
    os.makedirs(cfg_rootdir, exist_ok=True)
    config_file = os.path.join(cfg_rootdir, filename + ".ini")
    config = configparser.ConfigParser()
    if os.path.exists(config_file):
        config.read(config_file)
        return config, config_file
    config["DEFAULT"] = {"cube_size": str(cube_size), "min_num": "64", "resolution": str(resolution)}
    simple = mode == "factorized" and modelname.endswith("model_simple")
    rates = HYPER_RATES if mode == "hyper" else (FACTORIZED_RATES_SIMPLE if simple else FACTORIZED_RATES)
    root = os.path.join(ckpt_root, "simple") if simple else ckpt_root
    for name, scale, alpha in rates:
        short = os.path.join(root, "a%gb3" % alpha)
        long_ = os.path.join(root, "a%.2fb%.2f" % (alpha, 3.0))
        config[name] = {"scale": str(scale), "ckpt_dir": (short if os.path.isdir(short) or not os.path.isdir(long_) else long_) + "/"}
    with open(config_file, "w") as f:
        config.write(f)
    return config, config_file


def start_rate_point(points, model, ckpt_dir, scale, cube_size, min_num):
    """Partition + encode of one rate point started on a helper thread / stream (transform.compress_hyper_ahead): pass the
    result to rate_point(..., started=...).  eval() starts rate k + 1 before it decodes and measures rate k."""
    cubes, cube_positions, points_numbers = preprocess_points(points, scale, cube_size, min_num)
    return cube_positions, points_numbers, compress_hyper_ahead(cubes, model, ckpt_dir)


def _rate_point_factorized(points, model, ckpt_dir, scale, cube_size, min_num, rootdir=None):
    """eval.py:45-75 without the metrics: compress_factorized, the three-file container written and read back,
    decompress_factorized.  bpps like rate_point's, the hyper and head terms 0 (eval.py:69-71)."""
    cubes, cube_positions, points_numbers = preprocess_points(points, scale, cube_size, min_num)
    strings, min_v, max_v, shape = compress_factorized(cubes, model, ckpt_dir)
    own_tmp = rootdir is None
    rootdir = rootdir or tempfile.mkdtemp(prefix="pcgc_eval_")
    sizes = bs.write_binary_files_factorized("x", strings, points_numbers, cube_positions, min_v, max_v, shape, rootdir=rootdir,
                                             verbose=False)
    strings_d, nums_d, pos_d, min_v_d, max_v_d, shape_d = bs.read_binary_files_factorized("x", rootdir=rootdir)
    cubes_d = decompress_factorized(strings_d, min_v_d, max_v_d, shape_d, model, ckpt_dir)
    if own_tmp:
        for k in ("strings", "pointnums", "cubepos"):
            os.remove(os.path.join(rootdir, "x." + k))
        os.rmdir(rootdir)
    n = float(len(points))
    b_strings, b_nums, b_pos = sizes
    bpps = [round(8 * sum(sizes) / n, 4), round(8 * b_strings / n, 4), 0, 0, round(8 * b_nums / n, 4), round(8 * b_pos / n, 4)]
    return cubes_d, pos_d, nums_d, int(n), bpps


def rate_point(points, model, ckpt_dir, scale, cube_size, min_num, rootdir=None, started=None, mode="hyper"):
    """eval.py:77-113 (hyper) / 45-75 (factorized) without the metrics: returns (decoded cubes, cube_positions,
    points_numbers, N, bpps) with bpps = [total, strings, strings_hyper, strings_head, pointnums, cubepos] rounded to
    4 decimals like the reference."""
    if mode == "factorized":
        return _rate_point_factorized(points, model, ckpt_dir, scale, cube_size, min_num, rootdir)
    if started is not None:
        cube_positions, points_numbers, ahead = started
        stream = ahead.result()
    else:
        cubes, cube_positions, points_numbers = preprocess_points(points, scale, cube_size, min_num)
        stream = compress_hyper(cubes, model, ckpt_dir)
    y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape = stream
    own_tmp = rootdir is None
    rootdir = rootdir or tempfile.mkdtemp(prefix="pcgc_eval_")
    sizes = bs.write_binary_files_hyper("x", y_strings, z_string, points_numbers, cube_positions, y_min_vs, y_max_vs, y_shape,
                                        z_min_v, z_max_v, z_shape, rootdir=rootdir, verbose=False)
    r = bs.read_binary_files_hyper("x", rootdir=rootdir)
    cubes_d = decompress_hyper(r[0], r[4], r[5], r[6], r[1], r[7], r[8], r[9], model, ckpt_dir)
    if own_tmp:
        for k in ("strings", "strings_head", "strings_hyper", "pointnums", "cubepos"):
            os.remove(os.path.join(rootdir, "x." + k))
        os.rmdir(rootdir)
    n = float(len(points))
    b_strings, b_head, b_hyper, b_nums, b_pos = sizes
    bpps = [round(8 * sum(sizes) / n, 4)] + [round(8 * v / n, 4) for v in (b_strings, b_hyper, b_head, b_nums, b_pos)]
    return cubes_d, r[3], r[2], int(n), bpps


def eval(input_file, rootdir, cfgdir, res, mode="hyper", cube_size=64, modelname="pcgcv1_amd.models.model_voxception",
         fixed_thres=None, postfix=""):
    """eval.py:160-215.  The config .ini has DEFAULT {cube_size, min_num} and one section per rate with
    {scale, ckpt_dir, rho_d1, rho_d2} (eval.py:170-183).  Returns the list of result rows (dicts)."""
    if mode not in ("hyper", "factorized"):
        raise ValueError("eval: mode must be 'hyper' or 'factorized' (got %r)" % (mode,))
    hyper = mode == "hyper"
    model = importlib.import_module("pcgcv1_amd." + modelname if modelname.startswith("models.") else modelname)
    points, normals = iop.load_ply_normals(input_file)
    filename = os.path.split(input_file)[-1][:-4]
    os.makedirs(rootdir, exist_ok=True)
    config = configparser.ConfigParser()
    config.read(cfgdir)
    cube_size = config.getint("DEFAULT", "cube_size", fallback=cube_size)
    min_num = config.getint("DEFAULT", "min_num", fallback=64)
    res = config.getint("DEFAULT", "resolution", fallback=res)           # eval_ablation_studies.py:272
    rows = []
    rates = config.sections()
    search_log = []                                           # (item, ladder index, rho, PSNR) of every search step taken
    eval.last_search_log = search_log

    def start(rate):
        return start_rate_point(points, model, str(config.get(rate, "ckpt_dir")), float(config.get(rate, "scale")), cube_size, min_num)
    started = start(rates[0]) if rates and hyper else None
    for k, rate in enumerate(rates):
        scale = float(config.get(rate, "scale"))
        ckpt_dir = str(config.get(rate, "ckpt_dir"))
        if hyper:
            cur, started = started, None
            cur[2].result()                                   # this rate's strings exist (rate_point picks them up below)
            if k + 1 < len(rates):
                started = start(rates[k + 1])                 # the next rate's encode runs under this rate's decode + metrics
            cubes_d, cube_positions, points_numbers, n, bpps = rate_point(points, model, ckpt_dir, scale, cube_size, min_num, started=cur)
        else:                                                 # eval.py:188-189: test_factorized (one stream per cloud, nothing to overlap)
            cubes_d, cube_positions, points_numbers, n, bpps = rate_point(points, model, ckpt_dir, scale, cube_size, min_num,
                                                                          mode="factorized")

        def measure(rho):
            rec = postprocess_points(cubes_d, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres)
            if metrics._off_grid(rec):                        # scale != 1: pc_error measures the float coordinates (process.py:76-77)
                return metrics.pc_error(points, rec, normals, res - 1)
            rec = np.unique(np.rint(rec).astype(np.int32), axis=0)          # pc_error drops duplicate points (dropDuplicates 2)
            return metrics.pc_error(points, rec, normals, res - 1)

        cache = {}

        def measured(rho):                                    # the ladders revisit rho = 1.0 and the chosen values
            if rho not in cache:
                cache[rho] = measure(rho)
            return cache[rho]
        if fixed_thres is None:                               # eval_ablation_studies.py:285-291
            rho_d1, rho_d2 = cfg_post_process(config, cfgdir, rate, measured, have_normals=normals is not None, log=search_log)
        else:
            rho_d1, rho_d2 = 1.0, 1.0
        row = dict(measured(1.0))
        r1, r2 = measured(rho_d1), measured(rho_d2)
        row.update({"ori_points": n, "scale": scale, "bpp": bpps[0], "bpp_strings": bpps[1], "bpp_strings_hyper": bpps[2],
                    "bpp_strings_head": bpps[3], "bpp_pointsnums": bpps[4], "bpp_cubepos": bpps[5], "rho_d1": rho_d1,
                    "optimal D1 PSNR": r1["mseF,PSNR (p2point)"], "rho_d2": rho_d2,
                    "optimal D2 PSNR": r2.get("mseF,PSNR (p2plane)", float("nan")), "rate": rate})
        rows.append(row)
        with open(os.path.join(rootdir, filename + postfix + ".csv"), "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=list(rows[0]))
            wr.writeheader()
            wr.writerows(rows)
    return rows


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)          # eval.py:217-246
    ap.add_argument("--input", type=str, nargs="+", dest="input", required=True)
    ap.add_argument("--rootdir", type=str, default="results/hyper/")
    ap.add_argument("--cfgdir", type=str, default="results/hyper/8iVFB_vox10.ini")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--mode", type=str, default="hyper")
    ap.add_argument("--cube_size", type=int, default=64)
    ap.add_argument("--modelname", type=str, default="pcgcv1_amd.models.model_voxception")
    ap.add_argument("--fixed_thres", type=float, default=None)
    ap.add_argument("--postfix", type=str, default="")
    a = ap.parse_args(argv)
    for input_file in sorted(a.input):
        for r in eval(input_file, a.rootdir, a.cfgdir, a.res, a.mode, a.cube_size, a.modelname, a.fixed_thres, a.postfix):
            print(r)


if __name__ == "__main__":
    main()
