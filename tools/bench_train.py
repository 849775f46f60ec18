"""Timing of one train_hyper step (BASELINE config 4 shape: batch 8 cubes of 64^3 per GPU).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv1_amd import synthetic
from pcgcv1_amd.train_hyper import Trainer
w = synthetic.make_weights(seed=1300, profile="dense")
x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=8)).cuda()
tr = Trainer(w, alpha=0.75, beta=3.0, lr=1e-5)
tr.step(x)
torch.cuda.synchronize(); t = time.perf_counter()
n = 3
for _ in range(n):
    terms = tr.step(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("train step: %.1f ms for 8 cubes -> %.1f cubes/s ; loss %.4f" % (dt * 1e3, 8 / dt, terms["loss"]))
