"""Per-step times of the headline round trip (compress_hyper + decompress_hyper of the 205-cube cloud): is the spread between
bench runs made of a few slow steps (GC, allocator) or of uniformly slower steps (clocks)?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
def step():
    out = transform.compress_hyper(cubes, model, "bench")
    return transform.decompress_hyper(*out, model, "bench")
for _ in range(4): step()
torch.cuda.synchronize()
for label in ("gc on", "gc off", "gc on", "gc off"):
    if label == "gc off": gc.collect(); gc.disable()
    else: gc.enable()
    ts = []
    for i in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step()
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts)
    print("%-7s mean %.2f median %.2f min %.2f max %.2f  p90 %.2f  | %s" % (label, ts.mean(), np.median(ts), ts.min(), ts.max(), np.percentile(ts, 90), " ".join("%.1f" % t for t in ts)))
gc.enable()
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold(), "alloc stats: num_alloc_retries", torch.cuda.memory_stats().get("num_alloc_retries"), "reserved MB", torch.cuda.memory_reserved() >> 20)
