"""Timing of one train_hyper step (BASELINE config 4 shape: batch 8 cubes of 64^3 per GPU).  GPU box only.
    python tools/bench_train.py [steps]
Every step is bracketed by a device synchronisation; the first steps size the plan's partial-sum pool and torch's
allocator (80 and 25 ms); a full Python garbage collection over the set-up's objects used to cost one later step 50 ms
(the objects are frozen after the warm-up now, as the training driver does).  The figure is the MEDIAN of the timed steps,
printed with their mean and maximum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv1_amd import synthetic
from pcgcv1_amd.train_hyper import Trainer
w = synthetic.make_weights(seed=1300, profile="dense")
x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=8)).cuda()
tr = Trainer(w, alpha=0.75, beta=3.0, lr=1e-5)
for _ in range(3):
    tr.step(x)
import gc
gc.collect(); gc.freeze()        # the one 50 ms step of earlier runs was a full Python collection over the set-up's objects
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts = []
for _ in range(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    terms = tr.step(x)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
med = sorted(ts)[len(ts) // 2]
print("train step: %.2f ms for 8 cubes -> %.1f cubes/s (median of %d steps; mean %.2f ms, max %.1f ms) ; loss %.4f"
      % (med * 1e3, 8 / med, n, 1e3 * sum(ts) / n, 1e3 * max(ts), terms["loss"]))
