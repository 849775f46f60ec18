"""GPU box: cost of the sharded codec path on ONE rank (world 1, no process group) next to the plain path, with a
coarse breakdown — what a rank of an N-GPU job spends outside the collectives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pcgcv1_amd import checkpoint, process, sharding, synthetic, transform
from pcgcv1_amd.models import model_voxception as model

checkpoint._CACHE["bench"] = synthetic.make_weights(seed=1300, profile="sparse")
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
B = len(cubes)
ops = sharding.HipOps(model, "bench")


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


ms, out = t(lambda: transform.compress_hyper(cubes, model, "bench"))
print("compress_hyper            %.1f ms" % ms)
ms, _ = t(lambda: transform.decompress_hyper(*out, model, "bench"))
print("decompress_hyper          %.1f ms" % ms)
ms, blk = t(lambda: transform.compress_block(ops.c, cubes))
print("compress_block            %.1f ms" % ms)
ms, stream = t(lambda: sharding.compress_hyper_sharded(cubes, ops, total=B, points_numbers=nums))
print("compress_hyper_sharded    %.1f ms" % ms)
ms, zh = t(lambda: ops.decode_z(stream[4], stream[5], stream[6], stream[7]))
print("  decode_z                %.1f ms" % ms)
ms, lg = t(lambda: transform.decompress_block(ops.c, zh, stream[0], stream[1], stream[2], stream[3]))
print("  decompress_block        %.1f ms" % ms)
ms, mk = t(lambda: ops.classify(lg, nums, 1.0))
print("  classify (top-k)        %.1f ms" % ms)
ms, _ = t(lambda: sharding._pack_bits(mk))
print("  pack bits               %.1f ms" % ms)
ms, _ = t(lambda: sharding.decompress_hyper_sharded(stream[:8], ops, points_numbers=stream[8], packed=True))
print("decompress_hyper_sharded  %.1f ms" % ms)
if os.environ.get("PCGC_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        sharding.decompress_hyper_sharded(stream[:8], ops, points_numbers=stream[8], packed=True)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
