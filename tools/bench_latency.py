"""Latency of compress_hyper + decompress_hyper for small batches (serving-style calls), GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcgcv1_amd import synthetic, transform, checkpoint
from pcgcv1_amd.models import model_voxception as model

checkpoint._CACHE["bench"] = synthetic.make_weights(seed=1300, profile="sparse")
for B in (1, 2, 8, 32, 205):
    x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=B)).cuda()
    for _ in range(3):
        out = transform.compress_hyper(x, model, "bench"); transform.decompress_hyper(*out, model, "bench")
    torch.cuda.synchronize(); n = 10 if B < 100 else 5
    t = time.perf_counter()
    for _ in range(n):
        out = transform.compress_hyper(x, model, "bench")
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        xs = transform.decompress_hyper(*out, model, "bench")
    torch.cuda.synchronize(); td = (time.perf_counter() - t) / n
    print("B=%3d  encode %7.2f ms  decode %7.2f ms  -> %7.1f cubes/s" % (B, te * 1e3, td * 1e3, B / (te + td)))
