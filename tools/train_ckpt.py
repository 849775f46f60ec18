"""Train a hyperprior checkpoint with the repo's own Trainer on seeded synthetic surfaces and close the loop on it.

    python tools/train_ckpt.py --alpha 6 --beta 3 --minutes 20 --out gpurun_out/ckpt

1. data     seeded closed-surface clouds (synthetic.make_cloud, seeds 1..N, shell count / radii varied per seed) on a
            1024^3 grid -> the codec's own partition (min_num 64) -> 64^3 occupancy cubes kept in HBM as uint8;
            batches of 8 cubes drawn on the device with a random axis permutation + flips.  The evaluation cloud
            (seed 1300 = bench.py's cloud) is never part of the training set.
2. train    pcgcv1_amd.train_hyper.Trainer.step (train_hyper.py:174-214 of the reference), for a wall-clock budget;
            the learning rate is held for the first 70 % of the budget and decays geometrically to lr/20 after it.
3. save     TensorFlow tensor bundle (pcgcv1_amd/tf_bundle.py) under <out>/hyper/a<alpha>b<beta>/ with the model
            variables only, as the reference's checkpoints (train_hyper.py:107-111: the optimizer is not in the Checkpoint).
4. report   tools/eval_ckpt.py on the held-out cloud: estimated bits (sum of -log2 likelihood) vs the bytes the range
            coder writes, bpp, D1 / D2 PSNR -> <out>/report_a<alpha>b<beta>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cloud_params(seed):
    """Shape family of training cloud `seed`: shell count and radius range drawn from the seed."""
    r = np.random.default_rng(10_000 + seed)
    n_shells = int(r.integers(4, 10))
    rmin = float(r.uniform(0.03, 0.08))
    rmax = float(rmin + r.uniform(0.04, 0.12))
    return dict(n_shells=n_shells, rmin=rmin, rmax=rmax, oversample=2.0)


def build_dataset(n_clouds, cube_size=64, min_num=64, first_seed=1, log=print):
    import torch
    from pcgcv1_amd import process, synthetic
    parts, t0 = [], time.time()
    for seed in range(first_seed, first_seed + n_clouds):
        assert seed != 1300
        pts = synthetic.make_cloud(seed=seed, **cloud_params(seed))
        cubes, _, _ = process.preprocess_points(pts, 1.0, cube_size, min_num)
        parts.append(cubes.reshape(cubes.shape[:4]).to(torch.uint8))
    data = torch.cat(parts, 0)
    log("dataset: %d clouds -> %d cubes (%.1f MB in HBM, mean %.0f points per cube) in %.1f s"
        % (n_clouds, data.shape[0], data.numel() / 1e6, float(data.sum()) / data.shape[0], time.time() - t0))
    return data


def draw_batch(data, batch, gen):
    """8 random cubes with a random axis permutation and flips (the surfaces have no preferred orientation)."""
    import torch
    idx = torch.randint(0, data.shape[0], (batch,), device=data.device, generator=gen)
    x = data[idx]
    perm = torch.randperm(3, generator=gen, device=data.device).tolist()
    x = x.permute(0, 1 + perm[0], 1 + perm[1], 1 + perm[2])
    flips = [1 + a for a in range(3) if bool(torch.rand((), generator=gen, device=data.device) < 0.5)]
    if flips:
        x = x.flip(flips)
    return x.to(torch.float32).unsqueeze(-1).contiguous()


def save_only(weights, ckpt_dir, step):
    """checkpoint.save_tf, then drop the OLDER steps' bundle files: the new bundle and the `checkpoint` state file are
    complete before anything is deleted (a kill in between leaves two checkpoints, never none), and only files this
    function's own earlier calls can have written (ckpt-<other step>.index / .data-*) are removed."""
    import re
    from pcgcv1_amd import checkpoint
    prefix = checkpoint.save_tf(weights, ckpt_dir, step)
    keep = os.path.basename(prefix)
    for f in os.listdir(ckpt_dir):
        m = re.fullmatch(r"(ckpt-\d+)\.(index|data-\d{5}-of-\d{5})", f)
        if m and m.group(1) != keep and os.path.isfile(os.path.join(ckpt_dir, f)):
            os.remove(os.path.join(ckpt_dir, f))
    prune_state_file(ckpt_dir)
    return prefix


def prune_state_file(ckpt_dir):
    """The `checkpoint` state file lists only bundles that exist (tf.train.CheckpointManager and anything else that reads
    all_model_checkpoint_paths would otherwise meet a missing file; the reference's Saver keeps the two in step)."""
    import re
    path = os.path.join(ckpt_dir, "checkpoint")
    if not os.path.isfile(path):
        return
    out = []
    for line in open(path).read().splitlines():
        m = re.fullmatch(r'\s*all_model_checkpoint_paths:\s*"([^"]+)"\s*', line)
        if m and not os.path.isfile(os.path.join(ckpt_dir, m.group(1) + ".index")):
            continue
        out.append(line)
    with open(path + ".tmp", "w") as f:
        f.write("\n".join(out) + "\n")
    os.replace(path + ".tmp", path)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--alpha", type=float, default=6.0)
    ap.add_argument("--beta", type=float, default=3.0)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--minutes", type=float, default=20.0)
    ap.add_argument("--max_steps", type=int, default=0)
    ap.add_argument("--clouds", type=int, default=24)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--init", default="", help="checkpoint directory to start from (default: seeded He-scaled weights)")
    ap.add_argument("--out", default="gpurun_out/ckpt")
    ap.add_argument("--no_report", action="store_true")
    ap.add_argument("--save_minutes", type=float, default=5.0)
    a = ap.parse_args(argv)
    import torch
    from pcgcv1_amd import checkpoint, synthetic
    from pcgcv1_amd.train_hyper import Trainer
    tag = "a%.2fb%.2f" % (a.alpha, a.beta)
    ckpt_dir = os.path.join(a.out, "hyper", tag)
    os.makedirs(ckpt_dir, exist_ok=True)
    logf = open(os.path.join(a.out, "train_%s.log" % tag), "a")

    def log(s):
        print(s, flush=True)
        logf.write(s + "\n")
        logf.flush()
    log("== train_ckpt %s lr %g budget %.1f min" % (tag, a.lr, a.minutes))
    torch.cuda.set_device(0)
    data = build_dataset(a.clouds, log=log)
    weights = checkpoint.load(a.init) if a.init else synthetic.make_weights(seed=a.seed, profile="dense")
    tr = Trainer(weights, alpha=a.alpha, beta=a.beta, lr=a.lr)
    gen = torch.Generator(device=data.device)
    gen.manual_seed(1234 + a.seed)
    torch.manual_seed(4321 + a.seed)                 # the additive uniform noise of the step (torch.rand_like)
    import gc
    gc.collect()
    gc.freeze()
    budget = a.minutes * 60.0
    t0, acc, n_acc, curve, last_save = time.time(), {}, 0, [], 0.0
    while True:
        el = time.time() - t0
        if el >= budget or (a.max_steps and tr.t >= a.max_steps):
            break
        frac = el / budget
        tr.lr = a.lr if frac < 0.7 else a.lr * (0.05 ** ((frac - 0.7) / 0.3))
        terms = tr.step(draw_batch(data, a.batch_size, gen))
        if not np.isfinite(terms["loss"]):
            log("step %d: loss is not finite (%r) -- stopping" % (tr.t, terms))
            return 2
        for k in ("loss", "bpp_y", "bpp_z", "empty", "full"):
            acc[k] = acc.get(k, 0.0) + terms[k]
        n_acc += 1
        if a.save_minutes and el - last_save >= 60.0 * a.save_minutes:
            save_only(tr.weights(), ckpt_dir, tr.t)                # a timeout of the box keeps the latest one
            last_save = el
        if tr.t % 500 == 0:
            row = {k: v / n_acc for k, v in acc.items()}
            row.update(step=tr.t, minutes=round((time.time() - t0) / 60.0, 2), lr=tr.lr)
            curve.append(row)
            log("step %6d  loss %.4f  bpp_y %.4f  bpp_z %.4f  empty %.5f  full %.4f  lr %.2e  %.1f min"
                % (tr.t, row["loss"], row["bpp_y"], row["bpp_z"], row["empty"], row["full"], tr.lr, row["minutes"]))
            acc, n_acc = {}, 0
    torch.cuda.synchronize()
    dt = time.time() - t0
    log("trained %d steps in %.1f min (%.2f ms per step)" % (tr.t, dt / 60.0, 1e3 * dt / max(tr.t, 1)))
    w = tr.weights()
    prefix = save_only(w, ckpt_dir, tr.t)
    log("saved " + prefix)
    with open(os.path.join(a.out, "curve_%s.json" % tag), "w") as f:
        json.dump({"alpha": a.alpha, "beta": a.beta, "lr": a.lr, "steps": tr.t, "minutes": dt / 60.0, "batch_size": a.batch_size,
                   "clouds": a.clouds, "cubes": int(data.shape[0]), "curve": curve}, f)
    del tr, data
    torch.cuda.empty_cache()
    if not a.no_report:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import eval_ckpt
        rep = eval_ckpt.evaluate(ckpt_dir)
        rep["train"] = {"steps": curve[-1]["step"] if curve else 0, "minutes": round(dt / 60.0, 2), "lr": a.lr,
                        "final": curve[-1] if curve else None}
        with open(os.path.join(a.out, "report_%s.json" % tag), "w") as f:
            json.dump(rep, f, indent=1)
        log(json.dumps(rep))
    return 0


if __name__ == "__main__":
    sys.exit(main())
