"""compress_hyper of the bench cloud (two host pipelines, as in the headline) for several analysis chunk sizes; interleaved rounds, median per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
settings = sys.argv[1].split(":") if len(sys.argv) > 1 else ["16", "20", "26", "35", "52"]
ref = None
res = {s: [] for s in settings}
for rnd in range(4):
    for s in settings:
        os.environ["PCGC_CHUNKS_A"] = s + ",64,256"
        for _ in range(2): out = transform.compress_hyper(cubes, model, "bench")
        if ref is None: ref = out
        assert out[0] == ref[0] and out[4] == ref[4], "bytes changed with the chunk size"
        for _ in range(15):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            transform.compress_hyper(cubes, model, "bench")
            torch.cuda.synchronize(); res[s].append(1e3 * (time.perf_counter() - t0))
for s in settings:
    a = np.array(res[s])
    print("chunk %-4s compress_hyper median %.2f ms  mean %.2f  min %.2f  (n=%d)" % (s, np.median(a), a.mean(), a.min(), len(a)))
