"""Where do the slow round trips lose their time?  (python tools/exp/t_slow_steps.py [n] [enc|dec|rt])  One step in ten of the headline loop takes 4-10 ms longer than the median
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
def _cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("usage_usec", "nr_throttled", "throttled_usec")}
    except OSError:
        return {"usage_usec": 0, "nr_throttled": 0, "throttled_usec": 0}
LOG = []
_lib._trace = lambda label: LOG.append((time.perf_counter(), threading.current_thread().name, label))
runs = []
stats = []
for i in range(n):
    c0 = _cpu_stat()
    if which == "rt":                                         # the whole round trip, as the headline loop runs it
        torch.cuda.synchronize(); LOG.clear(); t0 = time.perf_counter()
        out = transform.compress_hyper(cubes, model, "bench")
        transform.decompress_hyper(*out, model, "bench")
        t_ret = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    elif which == "enc":
        torch.cuda.synchronize(); LOG.clear(); t0 = time.perf_counter()
        out = transform.compress_hyper(cubes, model, "bench")
        t_ret = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        transform.decompress_hyper(*out, model, "bench")
    else:
        out = transform.compress_hyper(cubes, model, "bench")
        torch.cuda.synchronize(); LOG.clear(); t0 = time.perf_counter()
        transform.decompress_hyper(*out, model, "bench")
        t_ret = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    c1 = _cpu_stat()
    stats.append((c1["nr_throttled"] - c0["nr_throttled"], (c1["throttled_usec"] - c0["throttled_usec"]) / 1e3, (c1["usage_usec"] - c0["usage_usec"]) / 1e3))
    runs.append((1e3 * (t1 - t0), 1e3 * (t_ret - t0), [(1e3 * (t - t0), th, lab) for t, th, lab in LOG]))
_lib._trace = None
ts = np.array([r[0] for r in runs])
print("%s: median %.2f ms, mean %.2f, p90 %.2f, max %.2f" % (which, np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max()))
order = np.argsort(ts)
med = order[len(order) // 2]
thr = [i for i, st in enumerate(stats) if st[0] > 0]
print("cgroup: %d of %d calls saw the group throttled (cpu.max %s); CPU per call: median %.0f ms, max %.0f ms" % (
    len(thr), n, open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?",
    np.median([st[2] for st in stats]), max(st[2] for st in stats)))
print("slowest ten calls: " + "  ".join("#%d %.1f ms (throttled %d x, %.1f ms; cpu %.0f ms)" % (k, ts[k], stats[k][0], stats[k][1], stats[k][2]) for k in order[::-1][:10]))
def show(k, title):
    total, ret, marks = runs[k]
    print("--- %s: call %d, %.2f ms (returned to the caller at %.2f)" % (title, k, total, ret))
    for t, th, lab in marks:
        print("   %8.2f  %-14s %s" % (t, th[-14:], lab))
show(med, "median")
for k in order[::-1][:4]:
    show(k, "slow")
