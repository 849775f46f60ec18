"""File-level multi-GPU sharding of compress_hyper / decompress_hyper.

The reference is single-GPU.  Cubes are independent units (transform.py:116-122, 157-168, 238-256: one cube
per call, zero-padded borders), so rank r of W takes the contiguous block of the (already key-sorted) cube
list `shard_range(B, r, W)` and runs the whole per-cube pipeline on its own GPU with no communication.
The format has exactly one cross-cube coupling, the hyperprior stream (entropy_model.py:249-259: ONE
min/max over all cubes and ONE range-coded string), which costs one small exchange:

  encode   all_reduce(MIN/MAX) of the local z-hat range (2 ints)
           gather to rank 0: z-hat symbols (int8, 4 KiB per cube), per-cube y strings + (min, max)
           rank 0 range-codes the single z string over the cubes in order
  decode   rank 0 decodes z (sequential, host), broadcasts z-hat and the header; every rank decodes and
           synthesises its block; the per-cube occupancy masks (after the on-GPU top-k) are gathered to rank 0

One process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The per-rank compute is injected (`ops`): `HipOps` wraps the MI355X codec (transform.Codec); the tests pass
an oracle-backed stand-in so the exchange logic is exercised with world_size 2 on CPU.
"""
import pickle

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block of rank `rank`: sizes differ by at most one, concatenation preserves order."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipOps(object):
    """Per-rank compute on the local MI355X through the C ABI (pcgcv1_amd.transform.Codec)."""

    def __init__(self, model, ckpt_dir):
        from . import transform
        self.c = transform.get_codec(model, ckpt_dir).require_hyper()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.lower_bound = transform.LOWER_BOUND

    def encode_local(self, cubes):
        c = self.c
        x = cubes if torch.is_tensor(cubes) else torch.from_numpy(np.ascontiguousarray(cubes, np.float32))
        ys = c.analysis_transform(x.to(self.device))
        zs = c.hyper_encoder(ys)
        z_hat, _ = c.entropy_bottleneck(zs, False)
        locs, scales = c.hyper_decoder(z_hat, lower_bound=self.lower_bound)
        y_strings, y_min, y_max = c.conditional_entropy_model.compress_cubes(ys, locs, scales)
        return z_hat.to(torch.int8).cpu().numpy(), y_strings, y_min, y_max, tuple(ys.shape[1:])

    def encode_z(self, z_hat_int, min_v, max_v):
        from . import coder_ops
        eb = self.c.entropy_bottleneck
        if max_v == min_v:
            max_v += 1
        cdf = eb._get_cdf(min_v, max_v)
        sym = (z_hat_int.reshape(-1, eb.channels).astype(np.int32) - min_v).astype(np.int16)
        return coder_ops.range_encode(sym, cdf), min_v, max_v

    def decode_z(self, z_string, min_v, max_v, z_shape):
        return self.c.entropy_bottleneck.decompress(z_string, min_v, max_v, z_shape).to(torch.int8).cpu().numpy()

    def decode_local(self, z_hat_int, y_strings, y_min, y_max, y_shape):
        c = self.c
        z = torch.from_numpy(z_hat_int.astype(np.float32)).to(self.device)
        locs, scales = c.hyper_decoder(z, lower_bound=self.lower_bound)
        ys = c.conditional_entropy_model.decompress_cubes(y_strings, locs, scales, y_min, y_max, y_shape)
        return c.synthesis_transform(ys)

    def classify(self, logits, points_numbers, rho):
        from .dataprocess import inout_points as iop
        return iop.select_voxels(logits, points_numbers, rho).cpu().numpy()


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _coll_device(group):
    """Tensors handed to collectives live where the backend wants them: HBM for RCCL ("nccl"), host for gloo."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_gather_bytes(payload, group, world):
    """Variable-length byte strings of every rank, via two plain all_gathers (sizes, then padded uint8 buffers) —
    only collectives every backend implements natively (RCCL has no gather of Python objects)."""
    dev = _coll_device(group)
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(payload):
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [bytes(b[:k].cpu().numpy()) for b, k in zip(bufs, sizes)]


def _gather_objects(obj, group, rank, world):
    """Python objects of all ranks, in rank order (rank 0 uses them; the exchange is symmetric)."""
    if world == 1:
        return [obj]
    parts = _all_gather_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL), group, world)
    return [pickle.loads(p) for p in parts] if rank == 0 else None


def _broadcast_object(obj, group, rank):
    dev = _coll_device(group)
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL) if rank == 0 else b""
    n = torch.tensor([len(data)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=0, group=group)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    dist.broadcast(buf, src=0, group=group)
    return obj if rank == 0 else pickle.loads(bytes(buf.cpu().numpy()))


def compress_hyper_sharded(cubes, ops, group=None):
    """All ranks call it with the SAME full cube list (or tensor); each encodes its block.
    Rank 0 returns the reference's tuple (y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v,
    z_shape); the other ranks return None."""
    rank, world = _world(group)
    B = len(cubes)
    lo, hi = shard_range(B, rank, world)
    z_hat, y_strings, y_min, y_max, y_tail = ops.encode_local(cubes[lo:hi])
    # global range of the hyperprior symbols: the only value every rank needs from the others
    mm = torch.tensor([int(z_hat.min()) if z_hat.size else 127, -(int(z_hat.max()) if z_hat.size else -128)],
                      dtype=torch.int32)
    if world > 1:
        mm = mm.to(_coll_device(group))
        dist.all_reduce(mm, op=dist.ReduceOp.MIN, group=group)
        mm = mm.cpu()
    z_min, z_max = int(mm[0]), -int(mm[1])
    parts = _gather_objects((z_hat, y_strings, np.asarray(y_min), np.asarray(y_max)), group, rank, world)
    if rank != 0:
        return None
    z_all = np.concatenate([p[0] for p in parts])
    ys = [s for p in parts for s in p[1]]
    y_min_vs = np.concatenate([p[2] for p in parts]).astype(np.int32)
    y_max_vs = np.concatenate([p[3] for p in parts]).astype(np.int32)
    z_string, z_min, z_max = ops.encode_z(z_all, z_min, z_max)
    return (ys, y_min_vs, y_max_vs, np.array((1,) + tuple(y_tail), np.int32), z_string, z_min, z_max,
            np.array(z_all.shape, np.int32))


def decompress_hyper_sharded(stream, ops, points_numbers=None, rho=1.0, group=None):
    """`stream` = the tuple compress_hyper returns (only rank 0's copy is read).  Returns on rank 0 either the
    logits of all cubes [B,cs,cs,cs,1] (points_numbers is None) or the uint8 occupancy masks after the
    per-cube top-k (the 32x smaller payload to exchange); None on the other ranks."""
    rank, world = _world(group)
    head = [None]
    if rank == 0:
        y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape = stream
        z_hat = ops.decode_z(z_string, z_min_v, z_max_v, z_shape)           # sequential by construction
        head = [(z_hat, list(y_strings), np.asarray(y_min_vs), np.asarray(y_max_vs), np.asarray(y_shape),
                 None if points_numbers is None else np.asarray(points_numbers))]
    if world > 1:
        head = [_broadcast_object(head[0], group, rank)]
    z_hat, y_strings, y_min_vs, y_max_vs, y_shape, nums = head[0]
    B = len(y_strings)
    lo, hi = shard_range(B, rank, world)
    logits = ops.decode_local(z_hat[lo:hi], y_strings[lo:hi], y_min_vs[lo:hi], y_max_vs[lo:hi], y_shape)
    if nums is None:
        payload = logits.cpu().numpy() if torch.is_tensor(logits) else np.asarray(logits)
    else:
        masks = np.asarray(ops.classify(logits, nums[lo:hi], rho), np.uint8)
        payload = (masks.shape, np.packbits(masks.reshape(-1)))            # 1 bit per voxel on the wire (32 KiB per 64^3 cube)
    parts = _gather_objects(payload, group, rank, world)
    if rank != 0:
        return None
    if nums is None:
        return np.concatenate(parts)
    return np.concatenate([np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape) for shape, bits in parts])
