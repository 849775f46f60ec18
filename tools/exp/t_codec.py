"""Median times of compress_hyper, decompress_hyper and the round trip on the bench cloud (two host pipelines, as in the headline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for _ in range(4):
    out = transform.compress_hyper(cubes, model, "bench"); transform.decompress_hyper(*out, model, "bench")
def _cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("usage_usec", "nr_throttled", "throttled_usec")}
    except OSError:
        return {}
c0, w0 = _cpu_stat(), time.perf_counter()
te, td = [], []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = transform.compress_hyper(cubes, model, "bench")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    transform.decompress_hyper(*out, model, "bench")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    te.append(1e3 * (t1 - t0)); td.append(1e3 * (t2 - t1))
te, td = np.array(te), np.array(td)
print("compress_hyper median %.2f ms (mean %.2f)   decompress_hyper median %.2f ms (mean %.2f)   round trip median %.2f mean %.2f" % (
    np.median(te), te.mean(), np.median(td), td.mean(), np.median(te + td), (te + td).mean()))
c1, w1 = _cpu_stat(), time.perf_counter()
if c0:
    print("  timed region: %.2f CPUs busy on average (cgroup usage / wall), throttled %d times, %.1f ms" % (
        (c1["usage_usec"] - c0["usage_usec"]) / 1e6 / (w1 - w0), c1["nr_throttled"] - c0["nr_throttled"], (c1["throttled_usec"] - c0["throttled_usec"]) / 1e3))
