"""Close the loop on a checkpoint: what the entropy models ESTIMATE for the held-out cloud against what the range coder
actually WRITES, and the rate / distortion point of the whole codec.

    python tools/eval_ckpt.py checkpoints/hyper/a6.00b3.00 [--seed 1300]

estimated bits = sum over the cloud of -log2 p(y_hat | loc, scale) (SymmetricConditional, training=False) and of
-log2 p(z_hat) (EntropyBottleneck, training=False) — the quantities train_hyper.py:193-196 of the reference optimises;
actual bytes   = the y strings (one per cube) + the single z string that compress_hyper returns.
The two differ by the CDF quantisation to 16 bits (every symbol of a cube's support gets >= 1/65536), the likelihood
bound, and the range coder's termination (<= 2 bytes + rounding per string: one string per cube); both parts are itemised.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def evaluate(ckpt_dir, seed=1300, chunk=64, cloud_kwargs=None):
    import torch
    from pcgcv1_amd import eval as rd
    from pcgcv1_amd import process, synthetic, transform
    from pcgcv1_amd.models import model_voxception as model
    pts = synthetic.make_cloud(seed=seed, **(cloud_kwargs or {}))
    cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
    B = int(cubes.shape[0])
    c = transform.get_codec(model, ckpt_dir).require_hyper()
    bits_y = bits_z = 0.0
    ymin, ymax, zmin, zmax = 0, 0, 0, 0
    zero_frac = 0.0
    for lo in range(0, B, chunk):
        x = cubes[lo:lo + chunk]
        ys = c.analysis_transform(x)
        zs = c.hyper_encoder(ys)
        z_hat, lik_z = c.entropy_bottleneck(zs, False)
        loc, scale = c.hyper_decoder(z_hat, lower_bound=transform.LOWER_BOUND)
        y_hat, lik_y = c.conditional_entropy_model(ys, loc, scale, False)
        bits_y += float(-torch.log2(lik_y.double()).sum())
        bits_z += float(-torch.log2(lik_z.double()).sum())
        ymin, ymax = min(ymin, int(y_hat.min())), max(ymax, int(y_hat.max()))
        zmin, zmax = min(zmin, int(z_hat.min())), max(zmax, int(z_hat.max()))
        zero_frac += float((y_hat == 0).double().mean()) * x.shape[0] / B
    try:
        out = transform.compress_hyper(cubes, model, ckpt_dir)
        xs = transform.decompress_hyper(*out, model, ckpt_dir)
        out2 = transform.compress_hyper(cubes, model, ckpt_dir, decompress=True)
    except Exception as e:                                 # noqa: BLE001 — e.g. an untrained model's symbol range exceeds the coder's 32
        return {"ckpt_dir": str(ckpt_dir), "error": repr(e), "estimated_bits": {"y": bits_y, "z": bits_z},
                "y_hat_range": [ymin, ymax], "z_hat_range": [zmin, zmax], "bpp_latents_estimated": (bits_y + bits_z) / len(pts)}
    same = bool(torch.equal(xs, out2[8]))                  # decoder == encoder-side reconstruction (eval.py:96-100 "cheat")
    # what an ideal arithmetic coder would spend with the 16-bit QUANTISED tables the range coder really uses
    # (pmf_to_quantized_cdf gives every symbol of a cube's support >= 1/65536, so a symbol the float model prices at
    # up to -log2(1e-9) = 29.9 bits costs at most 16; the float estimate and the coder can only be compared through it)
    from pcgcv1_amd import _lib
    lib = _lib.hip()
    cm = c.conditional_entropy_model
    qbits_y, rare = 0.0, 0
    mn_all, mx_all = np.asarray(out[1], np.int32), np.asarray(out[2], np.int32)
    for lo in range(0, B, chunk):
        x = cubes[lo:lo + chunk]
        nb = int(x.shape[0])
        ys = c.analysis_transform(x)
        z_hat, _ = c.entropy_bottleneck(c.hyper_encoder(ys), False)
        loc, scale = c.hyper_decoder(z_hat, lower_bound=transform.LOWER_BOUND)
        y_hat, lik_y = cm(ys, loc, scale, False)
        rare += int((lik_y < 2.0 ** -16).sum())
        mn_d = torch.from_numpy(mn_all[lo:lo + nb]).to(ys.device)
        mx_d = torch.from_numpy(mx_all[lo:lo + nb]).to(ys.device)
        rows = ys.numel()
        lohi = torch.empty(rows, dtype=torch.int32, device=ys.device)
        ncols = int((mx_all[lo:lo + nb] - mn_all[lo:lo + nb]).max()) + 1
        _lib.check(lib.pcgc_laplace_cdf(_lib.dptr(loc.reshape(-1)), _lib.dptr(scale.reshape(-1)), _lib.dptr(mn_d), _lib.dptr(mx_d),
                                        rows, rows // nb, ncols, 1e-9, _lib.dptr(y_hat.reshape(-1)), None, _lib.dptr(lohi),
                                        _lib.stream()), "pcgc_laplace_cdf")
        w = lohi.to(torch.int64)
        width = ((w >> 16) & 0xFFFF) + 1 - (w & 0xFFFF)
        qbits_y += float((16.0 - torch.log2(width.double())).sum())
    eb = c.entropy_bottleneck
    z_all = torch.cat([c.entropy_bottleneck(c.hyper_encoder(c.analysis_transform(cubes[lo:lo + chunk])), False)[0]
                       for lo in range(0, B, chunk)])
    zmn, zmx = int(out[5]), int(out[6])
    cdf_z = np.asarray(eb._get_cdf(zmn, zmx), np.int64).reshape(eb.channels, -1)
    sym = (z_all.reshape(-1, eb.channels).cpu().numpy().astype(np.int64) - zmn)
    ch = np.broadcast_to(np.arange(eb.channels), sym.shape)
    wz = cdf_z[ch, sym + 1] - cdf_z[ch, sym]
    qbits_z = float((16.0 - np.log2(wz.astype(np.float64))).sum())
    bytes_y, bytes_z = sum(len(s) for s in out[0]), len(out[4])
    npts = float(len(pts))
    try:
        r = rd.test_hyper(pts, model, ckpt_dir)            # container files + D1 (eval.py:77-113)
    except Exception as e:                                 # noqa: BLE001 — e.g. a symbol range the container cannot hold
        r = {"bpp": float("nan"), "d1_psnr": float("nan"), "n_points_in": len(pts), "n_points_out": -1, "error": repr(e)}
    est_bits, act_bits = bits_y + bits_z, 8.0 * (bytes_y + bytes_z)
    return {
        "ckpt_dir": str(ckpt_dir), "cloud": "synthetic.make_cloud(seed=%d): %d points, %d cubes of 64^3" % (seed, len(pts), B),
        "estimated_bits": {"y": round(bits_y, 1), "z": round(bits_z, 1), "total": round(est_bits, 1)},
        "actual_bytes": {"y_strings": bytes_y, "z_string": bytes_z, "total": bytes_y + bytes_z},
        "actual_over_estimated": round(act_bits / max(est_bits, 1e-9), 5),
        "actual_over_estimated_y": round(8.0 * bytes_y / max(bits_y, 1e-9), 5), "actual_over_estimated_z": round(8.0 * bytes_z / max(bits_z, 1e-9), 5),
        "quantised_table_bits": {"y": round(qbits_y, 1), "z": round(qbits_z, 1), "total": round(qbits_y + qbits_z, 1),
                                 "what": "sum of -log2((cdf[s+1]-cdf[s])/65536) with the 16-bit tables the range coder uses"},
        "actual_over_quantised_tables": round(act_bits / max(qbits_y + qbits_z, 1e-9), 5),
        "y_symbols_priced_below_2^-16_by_the_float_model": rare,
        "excess_bytes_per_string_over_tables": round((act_bits - qbits_y - qbits_z) / 8.0 / (B + 1), 3),
        "bytes_per_cube": round((bytes_y + bytes_z) / B, 2),
        "bpp_latents_estimated": round(est_bits / npts, 5), "bpp_latents_actual": round(act_bits / npts, 5),
        "bpp_files": round(r["bpp"], 5), "bpp_items": {k[4:]: round(v, 5) for k, v in r.items() if k.startswith("bpp_")},
        "d1_psnr_db": round(r["d1_psnr"], 4), "peak": 1023, "points_in": r["n_points_in"], "points_out": r["n_points_out"],
        "y_hat_range": [ymin, ymax], "z_hat_range": [zmin, zmax], "y_hat_zero_fraction": round(zero_frac, 5),
        "per_cube_y_range": [int(np.min(out[1])), int(np.max(out[2]))],
        "decoder_equals_encoder_side_reconstruction": same, "error": r.get("error"),
        "reference_recorded": {"where": "demo.ipynb:835-837, 922-924 (longdress_vox10_1300, hyper a0.75b3, trained on ShapeNet)",
                               "bpp": 0.1133, "d1_psnr_db": 67.7148, "bytes_per_cube": round((7128 + 4110) / 202.0, 1)},
    }


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt_dir")
    ap.add_argument("--seed", type=int, default=1300)
    a = ap.parse_args()
    print(json.dumps(evaluate(a.ckpt_dir, a.seed), indent=1))
