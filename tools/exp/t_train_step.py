"""Median train_hyper step (batch 8 x 64^3, as bench.py's train block) — for A/B of Trainer settings through the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import synthetic
from pcgcv1_amd.train_hyper import Trainer
tr = Trainer(synthetic.make_weights(seed=1300, profile="dense"), alpha=0.75, beta=3.0, lr=1e-5)
x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=8)).cuda()
for _ in range(4): tr.step(x)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    terms = tr.step(x)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts)
print("train step median %.3f ms  mean %.3f  min %.3f   loss %.6f  grad checksum %.9e" % (np.median(ts), ts.mean(), ts.min(), terms["loss"], float(tr.flat_g.double().abs().sum())))
