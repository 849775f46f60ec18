"""Where do the slow round trips lose their time?  One step in ten of the headline loop takes 4-10 ms longer than the median
(tools/exp/t_step_jitter.py).  Runs N synchronised compress_hyper calls (and decompress_hyper) with the _lib.mark hooks on, then prints,
for the slowest calls and for a median one, the time of every mark relative to the call's start, per thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import _lib, checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
which = sys.argv[2] if len(sys.argv) > 2 else "enc"
for _ in range(4):
    out = transform.compress_hyper(cubes, model, "bench"); transform.decompress_hyper(*out, model, "bench")
LOG = []
_lib._trace = lambda label: LOG.append((time.perf_counter(), threading.current_thread().name, label))
runs = []
for i in range(n):
    if which == "enc":
        torch.cuda.synchronize(); LOG.clear(); t0 = time.perf_counter()
        out = transform.compress_hyper(cubes, model, "bench")
        t_ret = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        transform.decompress_hyper(*out, model, "bench")
    else:
        out = transform.compress_hyper(cubes, model, "bench")
        torch.cuda.synchronize(); LOG.clear(); t0 = time.perf_counter()
        transform.decompress_hyper(*out, model, "bench")
        t_ret = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    runs.append((1e3 * (t1 - t0), 1e3 * (t_ret - t0), [(1e3 * (t - t0), th, lab) for t, th, lab in LOG]))
_lib._trace = None
ts = np.array([r[0] for r in runs])
print("%s: median %.2f ms, mean %.2f, p90 %.2f, max %.2f" % (which, np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max()))
order = np.argsort(ts)
med = order[len(order) // 2]
def show(k, title):
    total, ret, marks = runs[k]
    print("--- %s: call %d, %.2f ms (returned to the caller at %.2f)" % (title, k, total, ret))
    for t, th, lab in marks:
        print("   %8.2f  %-14s %s" % (t, th[-14:], lab))
show(med, "median")
for k in order[::-1][:3]:
    show(k, "slow")
