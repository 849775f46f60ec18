"""Worker for tests/test_gpu_sharding.py::test_data_parallel_step: one rank of a data-parallel train_hyper job
(Trainer.step: ONE all_reduce of the flat gradient buffer).  Usage: rank world port outfile backend"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pcgcv1_amd import synthetic                                           # noqa: E402
from pcgcv1_amd.train_hyper import Trainer                                 # noqa: E402


def main():
    rank, world, port, outfile, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    torch.cuda.set_device(rank % torch.cuda.device_count())
    if world > 1:
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    w = synthetic.make_weights(seed=5, profile="dense")
    w["hyper_decoder/conv4_2/bias"] = (w["hyper_decoder/conv4_2/bias"] + 0.8).astype(np.float32)
    B, cs = 2, 16
    # every rank gets its own batch (seeded by rank); world 1 runs rank 0's and rank 1's batches in turn
    def batch(r):
        x = synthetic.make_cubes(seed=30 + r, n_cubes=B, cube_size=cs, occupancy=0.06)
        rng = np.random.default_rng(40 + r)
        ny = (rng.random((B, cs // 4, cs // 4, cs // 4, 16)) - 0.5).astype(np.float32)
        nz = (rng.random((B, cs // 8, cs // 8, cs // 8, 8)) - 0.5).astype(np.float32)
        return x, ny, nz
    tr = Trainer(w, alpha=0.75, beta=3.0, lr=1e-3)
    if world > 1:
        for _ in range(2):
            tr.step(*batch(rank))
        out = {"weights": tr.weights()}
    else:
        # reference for world 2: mean of the two replica gradients, then the same Adam update
        for _ in range(2):
            tr.forward_backward(*batch(0), grad_scale=0.5)
            g0 = tr.flat_g.clone()
            tr.forward_backward(*batch(1), grad_scale=0.5)
            tr.flat_g.add_(g0)
            tr.apply_gradients()
        out = {"weights": tr.weights()}
    if rank == 0:
        with open(outfile, "wb") as f:
            pickle.dump(out, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
