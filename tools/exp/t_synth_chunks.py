"""Synthesis of the 205-cube cloud for several 64^3 chunk sizes: total time and per-kernel time per cube (profiled, serial)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
c = transform.get_codec(model, "bench")
y = torch.round(c.analysis_transform(cubes))
net = c.synthesis_transform
for nb in [int(v) for v in os.environ.get("NBS", "205,79").split(",")]:
  for env in os.environ.get("CHUNKS", "8,64,256:16,64,256").split(":"):
    os.environ["PCGC_CHUNKS_S"] = env
    yy = y[:nb].contiguous()
    for _ in range(2): net(yy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): net(yy)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 5
    net.set_profiling(True); net(yy); rows = net.profile_report(); net.set_profiling(False)
    agg = {}
    for r in rows:
        k = "%s@%d" % (r["kernel"], r["Din"])
        agg[k] = agg.get(k, 0.0) + r["ms"]
    print("B=%d chunks %-12s synthesis %.2f ms (%.1f us per cube); " % (nb, env, tot, 1e3 * tot / nb) + "  ".join("%s %.1f" % (k, 1e3 * v / nb) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:14]) + "  [us per cube]")
