import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import synthetic, checkpoint, transform
from pcgcv1_amd.models import model_voxception as model
checkpoint._CACHE["t"] = synthetic.make_weights(seed=3, profile="sparse")
c = transform.get_codec(model, "t")
x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=64, cube_size=64)).cuda()
y = torch.randn((64, 16, 16, 16, 16), device="cuda")
for net, inp in ((c.analysis_transform, x), (c.synthesis_transform, y)):
    for _ in range(2):
        net(inp)
    net.set_profiling(True)
    for _ in range(3):
        net(inp)
    torch.cuda.synchronize()
    agg = {}
    for r in net.profile_report():
        if r["kernel"] in ("rowup", "rowdown"):
            a = agg.setdefault(r["kernel"], [0.0, 0]); a[0] += r["ms"]; a[1] += 1
    net.set_profiling(False)
    for k, v in agg.items():
        print(k, "avg us per launch", 1e3 * v[0] / v[1], "launches", v[1])
