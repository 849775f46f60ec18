"""Worker for tests/test_gpu_sharding.py: one rank of a world_size-N job running the sharded codec with the
real per-rank compute (sharding.HipOps, HIP kernels through the C ABI).  On the 1-GPU test box all ranks share
cuda:0 and rendezvous over gloo; on an 8-GPU node the same code runs with backend nccl and one GPU per rank.
Usage: rank world port outfile backend n_cubes"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pcgcv1_amd import sharding, synthetic                                # noqa: E402
from pcgcv1_amd.models import model_voxception as model                   # noqa: E402


def main():
    rank, world, port, outfile, backend, n_cubes = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                                    sys.argv[5], int(sys.argv[6]))
    torch.cuda.set_device(rank % torch.cuda.device_count())
    if world > 1 or backend == "nccl":      # world 1 + nccl: every collective still goes through RCCL on device buffers
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    if n_cubes == 0:
        # the bench's own shape: the 205-cube cloud at 64^3 — every rank's block (>= 96 cubes) runs as two host pipelines
        # with the early-z hook on a pipeline thread, on the row kernels bench.py --gpus N times
        from pcgcv1_amd import process
        cubes, _, nums = process.preprocess_points(synthetic.make_cloud(seed=1300), 1.0, 64, 64)
        n_cubes = int(cubes.shape[0])
        ops = sharding.HipOps(model, "synthetic:1300:sparse")
        lo, hi = sharding.shard_range(n_cubes, rank, world)
        ex = sharding.Exchange(timing=True)
        stream = sharding.compress_hyper_sharded(cubes[lo:hi].contiguous(), ops, total=n_cubes, points_numbers=nums[lo:hi], exchange=ex)
        path = dict(ops.c.last_path)
        ex_d, ex_a = sharding.Exchange(), sharding.Exchange()
        masks = sharding.decompress_hyper_sharded(stream[:8] if rank == 0 else None, ops, points_numbers=stream[8] if rank == 0 else None,
                                                  exchange=ex_d)
        # the all-gather form: every rank ends up with the whole cloud's masks (all_gather_into_tensor on the backend's buffers)
        masks_all = sharding.decompress_hyper_sharded(stream[:8] if rank == 0 else None, ops,
                                                      points_numbers=stream[8] if rank == 0 else None, exchange=ex_a, gather_all=True)
        assert masks_all is not None and masks_all.shape == (n_cubes, 64, 64, 64, 1)
        where = {"nccl": "cuda", "gloo": "cpu"}[backend] if dist.is_initialized() else "cpu"
        assert ex.device.type == where and ex_d.device.type == where
        if rank == 0:
            assert np.array_equal(masks, masks_all)
            with open(outfile, "wb") as f:
                pickle.dump({"stream": stream, "masks_packed": np.packbits(masks.reshape(n_cubes, -1), axis=1), "path": path,
                             "collectives": [c[0] for c in ex.log], "collectives_decode": [c[0] for c in ex_d.log],
                             "collectives_gather_all": [c[0] for c in ex_a.log], "bytes": [c[1] for c in ex.log + ex_d.log],
                             "early_z": bool(ops.early_z), "collective_device": ex.device.type}, f)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    cubes = synthetic.make_cubes(seed=9, n_cubes=n_cubes, cube_size=32, occupancy=0.03)
    nums = cubes.sum(axis=(1, 2, 3, 4)).astype(np.uint16)
    ops = sharding.HipOps(model, "synthetic:21:dense")
    stream = sharding.compress_hyper_sharded(cubes, ops)
    lo, hi = sharding.shard_range(n_cubes, rank, world)
    ex = sharding.Exchange(timing=True)
    local = sharding.compress_hyper_sharded(cubes[lo:hi], ops, total=n_cubes, points_numbers=nums[lo:hi], exchange=ex)
    if rank == 0:       # block-local form (what test.py uses): same stream, point counts gathered with it
        assert list(local[0]) == list(stream[0]) and local[4] == stream[4] and np.array_equal(local[8], nums)
        assert not dist.is_initialized() or [c[0] for c in ex.log][1:] == ["gather z-hat", "all_reduce y bytes", "gather per-cube records", "gather y strings"]
    logits = sharding.decompress_hyper_sharded(stream, ops)
    masks = sharding.decompress_hyper_sharded(stream, ops, points_numbers=nums, rho=1.0)
    if rank == 0:
        with open(outfile, "wb") as f:
            pickle.dump({"stream": stream, "logits": logits, "masks": masks}, f)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
