"""File-level multi-GPU sharding of compress_hyper / decompress_hyper.

The reference is single-GPU.  Cubes are independent units (transform.py:116-122, 157-168, 238-256: one cube
per call, zero-padded borders), so rank r of W takes the contiguous block `shard_range(B, r, W)` of the
(already key-sorted) cube list and runs the whole per-cube pipeline on its own GPU with no communication.
The format has exactly one cross-cube coupling, the hyperprior stream (entropy_model.py:249-259: ONE
min/max over all cubes and ONE range-coded string).  Everything that crosses ranks is a plain tensor
collective on a pre-sized buffer — what RCCL implements natively over xGMI; no Python objects travel:

  encode   all_reduce(MIN)            int32[3]            range of the z-hat symbols + status word
           gather (to rank 0)         int8 [W, bmax*zlen] z-hat symbols (4 KiB per 64^3 cube)
           all_reduce(MIN)            int32[2]            -(bytes of the largest block of y strings) = the gather buffer size, status
           gather (to rank 0)         int32[W, bmax*4]    per cube: string length, y min, y max, point count
           gather (to rank 0)         uint8[W, cap]       the ranks' concatenated y strings
           rank 0 range-codes the single z string over the cubes in order (sequential host tail)
  decode   broadcast                  int64[16]           header (B, bytes, shapes, z range)
           broadcast                  uint8[len]          the z string: every rank decodes its own prefix of it
           broadcast                  int32[B*4]          per cube: string length, y min, y max, point count
           broadcast                  uint8[total]        y strings
           all_reduce(MIN)            int32[1]            decode status (a rank that failed locally says so before the gather)
           gather (to rank 0)         uint8[W, bmax*vox/8] bit-packed occupancy masks after the on-GPU top-k
                                      (or float32 logits when no point counts are given); all_gather_into_tensor
                                      with gather_all=True

Encoder blocks differ by at most one cube (buffers padded to bmax = ceil(B / W) cubes per rank); decoder blocks shrink
geometrically with the rank (`decode_ranges`: later ranks wait longer for their z symbols and get fewer cubes).  Tensors handed to
a collective live in HBM under "nccl" (= RCCL) and on the host under "gloo" (CPU tests).  `Exchange.log` keeps
bytes and wall time per collective (bench.py reports them).  The per-rank compute is injected (`ops`): `HipOps`
wraps the MI355X codec; the CPU tests pass an oracle-backed stand-in, so the exchange logic runs with world 2 on gloo.
"""
import time

import threading

import numpy as np
import torch
import torch.distributed as dist


class PeerFailure(RuntimeError):
    """Another rank failed in its local part of a sharded call.  The failing rank still takes part in the NEXT collective
    of the protocol with a poisoned status word, so every rank learns of the failure there, leaves the call with an
    exception (the failing rank with its own, the others with this one) and nobody is left waiting inside a collective."""


_I32_MAX = 2 ** 31 - 1


def shard_range(n, rank, world):
    """Contiguous block of rank `rank`: sizes differ by at most one, concatenation preserves order."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# The decoder's one sequential item is the z string (entropy_model.py:287-306: one range-coded stream over all cubes).
# Every rank decodes it itself — only up to the end of its own block — so rank r can start after the time the first
# hi_r cubes' symbols take, not after the whole string plus a broadcast.  Later ranks start later, so they get FEWER
# cubes: with z decoding costing tau per cube and the rest of a rank's decode c per cube, all ranks finish together when
# the block sizes shrink by rho = 1 / (1 + tau / c) from one rank to the next.  Measured on MI355X (205-cube batch):
# tau = 3.3 ms / 205, c = 28 ms / 205 -> tau / c = 0.118, rho = 0.894.
_DECODE_RHO = float(__import__("os").environ.get("PCGC_DECODE_RHO", "0.894"))


def decode_ranges(n, world, rho=None):
    """Contiguous decoder blocks [(lo, hi)] per rank with geometrically shrinking sizes (see above); rho = 1 gives
    shard_range's equal blocks.  Sizes sum to n, are non-increasing, and concatenation preserves order."""
    rho = _DECODE_RHO if rho is None else rho
    w = np.power(float(rho), np.arange(world))
    ideal = n * w / w.sum()
    sizes = np.floor(ideal).astype(np.int64)
    for i in np.argsort(-(ideal - sizes), kind="stable")[:int(n - sizes.sum())]:     # largest remainders get the rest
        sizes[i] += 1
    sizes = -np.sort(-sizes, kind="stable")
    hi = np.cumsum(sizes)
    return [(int(h - s_), int(h)) for h, s_ in zip(hi, sizes)]


class HipOps(object):
    """Per-rank compute on the local MI355X through the C ABI (pcgcv1_amd.transform.Codec)."""

    def __init__(self, model, ckpt_dir):
        from . import transform
        self.c = transform.get_codec(model, ckpt_dir).require_hyper()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.lower_bound = transform.LOWER_BOUND

    # encode_local takes z_hook and calls it before the y strings are coded (the z leg's collectives are then issued from a
    # pipeline thread, in the same order on every rank); PCGC_EARLY_Z=0 issues every collective from the calling thread
    early_z = __import__("os").environ.get("PCGC_EARLY_Z", "1") != "0"

    def encode_local(self, cubes, z_hook=None):
        """-> (z_hat float [b,...] on the device, y_strings, y_min, y_max, shape of one cube's y)."""
        from . import transform
        return transform.compress_block(self.c, cubes, z_hook)

    def encode_z(self, z_hat_int, min_v, max_v):
        from . import coder_ops
        eb = self.c.entropy_bottleneck
        if max_v == min_v:
            max_v += 1
        cdf = eb._get_cdf(min_v, max_v)
        z = np.asarray(z_hat_int)
        if z.dtype not in (np.int8, np.int16):
            z = z.astype(np.int16)
        return coder_ops.range_encode_values(z.reshape(-1, eb.channels), min_v, cdf), min_v, max_v

    def decode_z(self, z_string, min_v, max_v, z_shape):
        return self.c.entropy_bottleneck.decompress(z_string, min_v, max_v, z_shape)

    def decode_local(self, z_hat, y_strings, y_min, y_max, y_shape):
        from . import transform
        return transform.decompress_block(self.c, z_hat, y_strings, y_min, y_max, y_shape)

    def classify(self, logits, points_numbers, rho):
        from .dataprocess import inout_points as iop
        return iop.select_voxels(logits, points_numbers, rho)


class Exchange(object):
    """The collectives of one process group, on the device the backend wants, with a (name, bytes, ms) log."""

    def __init__(self, group=None, timing=False):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.nccl = self.on and dist.get_backend(group) == "nccl"
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.nccl else torch.device("cpu")
        self.timing = timing
        self.log = []

    def put(self, t):
        """tensor / ndarray -> contiguous tensor on the collective device"""
        t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))
        return t.to(self.device).contiguous()

    def _run(self, name, nbytes, fn):
        if not self.on:                  # no process group: a single process holds everything already
            return
        if self.timing:
            if self.nccl:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
        fn()
        if self.timing:
            if self.nccl:
                torch.cuda.synchronize()
            self.log.append((name, int(nbytes), 1e3 * (time.perf_counter() - t0)))
        else:
            self.log.append((name, int(nbytes), None))

    def all_reduce_min(self, name, t):
        t = self.put(t)
        self._run(name, t.numel() * t.element_size(), lambda: dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group))
        return t

    def all_gather(self, name, t):
        """-> [world, t.numel()] (every rank passes the same number of elements)"""
        t = self.put(t).reshape(-1)
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=self.device)
        if not self.on:
            out[0] = t
            return out
        self._run(name, out.numel() * out.element_size(), lambda: dist.all_gather_into_tensor(out.reshape(-1), t, group=self.group))
        return out

    def gather(self, name, t):
        """-> [world, t.numel()] on rank 0, None elsewhere (every rank passes the same number of elements)"""
        t = self.put(t).reshape(-1)
        if not self.on:
            return t.reshape(1, -1).clone()
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=self.device) if self.rank == 0 else None
        parts = [out[r] for r in range(self.world)] if self.rank == 0 else None
        self._run(name, self.world * t.numel() * t.element_size(), lambda: dist.gather(t, parts, dst=0, group=self.group))
        return out

    def broadcast(self, name, t):
        t = self.put(t)
        self._run(name, t.numel() * t.element_size(), lambda: dist.broadcast(t, src=0, group=self.group))
        return t


def _fill(buf, t):
    """copy `t` (any device) into the head of the pre-allocated flat buffer `buf`"""
    t = t.reshape(-1)
    buf[:t.numel()].copy_(t)


def _pad_to(t, n):
    t = t.reshape(-1)
    if t.numel() == n:
        return t
    out = torch.zeros(n, dtype=t.dtype, device=t.device)
    out[:t.numel()] = t
    return out


def _bytes_tensor(strings):
    data = b"".join(bytes(s) for s in strings)
    return torch.frombuffer(bytearray(data), dtype=torch.uint8) if data else torch.zeros(0, dtype=torch.uint8)


def compress_hyper_sharded(cubes, ops, group=None, total=None, points_numbers=None, exchange=None):
    """Every rank encodes its block of the cube list.

    cubes           the whole list / tensor (each rank slices its block), or — with `total` = number of cubes of the
                    whole cloud — only this rank's block shard_range(total, rank, world) (so that a rank voxelises and
                    uploads nothing but its own cubes).
    points_numbers  optional per-cube point counts of THIS rank's block; they ride along in the per-cube record and
                    come back for all cubes as a ninth tuple element on rank 0 (test.py writes them to .pointnums).
    Rank 0 returns the reference's tuple (y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape)
    (+ points_numbers when given); the other ranks return None."""
    ex = exchange or Exchange(group)
    rank, world = ex.rank, ex.world
    if total is None:
        B = len(cubes)
        lo, hi = shard_range(B, rank, world)
        cubes = cubes[lo:hi]
    else:
        B = int(total)
        lo, hi = shard_range(B, rank, world)
        assert len(cubes) == hi - lo, "rank %d holds %d cubes, its block has %d" % (rank, len(cubes), hi - lo)
    nb, bmax = hi - lo, -(-B // world)
    zbox = {}

    def exchange_z(z_hat):
        """The z leg: global symbol range, every rank's z-hat to every rank, and on rank 0 the single z string over the
        cubes of ALL ranks on a host thread.  With ops.early_z this runs from the pipeline thread that sees the block's
        hyper-latents first, while the y strings are still being coded — the serial z coding (2 ms per 205 cubes, times the
        number of ranks) leaves the step's critical path."""
        z_hat = z_hat if torch.is_tensor(z_hat) else torch.from_numpy(np.asarray(z_hat))
        z_tail = tuple(int(v) for v in z_hat.shape[1:])
        zlen = int(np.prod(z_tail))
        # global range of the hyperprior symbols, taken BEFORE the int8 cast: the only value every rank needs from the others
        zmn, zmx = (int(z_hat.min()), int(z_hat.max())) if nb else (_I32_MAX, -_I32_MAX)
        # the gather's buffer is made BEFORE the first collective: a failure to allocate it is then a failure "before the z
        # leg" (the handler below enters the all_reduce with status -1), and a failure while it is FILLED (after the
        # all_reduce) cannot take the buffer away — the gather is always entered with memory that exists
        zbuf = torch.zeros(bmax * zlen, dtype=torch.int8, device=ex.device)
        zbox["entered"] = True
        mm = ex.all_reduce_min("all_reduce z range", torch.tensor([zmn, -zmx, 0], dtype=torch.int32)).cpu()
        zbox["range_done"] = True
        # The two raises below are VERDICTS OF THE COLLECTIVE: every rank sees the same words and raises at the same point,
        # so nobody enters another collective.  The flag (not the exception's type) tells the caller's handler so: a purely
        # local OverflowError later on (a y symbol range the coder refuses) must still carry its status into the next
        # collective, and a local error that happens to surface first on a rank whose z leg got the verdict must not.
        if int(mm[2]) < 0:
            zbox["collective_verdict"] = True
            raise PeerFailure("a peer rank failed before the z leg of compress_hyper_sharded")
        z_min, z_max = int(mm[0]), -int(mm[1])
        if z_min < -128 or z_max > 127:
            zbox["collective_verdict"] = True
            raise OverflowError("hyperprior symbols %d..%d do not fit the container's int8 range (inout_bitstream.py:104-105)"
                                % (z_min, z_max))
        # a LOCAL failure while the gather's buffer is filled still enters the gather (the peers are in it) with the
        # pre-allocated buffer; the status then rides in the next all_reduce like any other local failure after the z leg
        try:
            _fill(zbuf, z_hat.to(torch.int8))
        except BaseException as e:                             # noqa: BLE001 — re-raised right after the gather
            zbox["local_error"] = e
        z_all = ex.gather("gather z-hat", zbuf)
        if "local_error" in zbox:
            raise zbox["local_error"]
        zbox["tail"] = z_tail
        if rank != 0:
            return
        z_np = z_all.cpu().numpy().reshape(world, bmax, zlen)
        z_cat = np.concatenate([z_np[r, :shard_range(B, r, world)[1] - shard_range(B, r, world)[0]] for r in range(world)])
        z_cat = z_cat.reshape((B,) + z_tail)
        zbox["shape"] = z_cat.shape

        dev = z_all.device if z_all.is_cuda else None

        def code():
            try:
                if dev is not None:                            # a fresh thread starts on device 0 (the CDF table may be built here)
                    from . import _lib
                    _lib.bind_device(dev)
                zbox["coded"] = ops.encode_z(z_cat, z_min, z_max)
            except BaseException as e:                         # noqa: BLE001 (re-raised by the caller's join)
                zbox["error"] = e
        zbox["thread"] = threading.Thread(target=code, name="pcgc-z-string")
        zbox["thread"].start()

    # A LOCAL failure (a kernel error, a symbol range the coder refuses, ...) must not leave the peers waiting in the next
    # collective: the failing rank enters that collective with status -1 in the word every all_reduce(MIN) of the protocol
    # carries, then re-raises; the peers raise PeerFailure right after the same collective.
    err = None
    rec = np.zeros((bmax, 4), np.int32)
    try:
        if getattr(ops, "early_z", False):
            z_hat, y_strings, y_min, y_max, y_tail = ops.encode_local(cubes, exchange_z)
        else:
            z_hat, y_strings, y_min, y_max, y_tail = ops.encode_local(cubes)
            exchange_z(z_hat)
        rec[:nb, 0] = [len(s) for s in y_strings]
        rec[:nb, 1], rec[:nb, 2] = np.asarray(y_min), np.asarray(y_max)
        if points_numbers is not None:
            rec[:nb, 3] = np.asarray(points_numbers)
    except BaseException as e:                                 # noqa: BLE001 — re-raised below, after the peers know
        err = e
    if err is not None:
        if zbox.get("collective_verdict"):
            raise err                                          # every rank leaves together, right after the z all_reduce
        if not zbox.get("entered"):                            # this rank failed before its z leg: the peers wait there
            ex.all_reduce_min("all_reduce z range", torch.tensor([_I32_MAX, _I32_MAX, -1], dtype=torch.int32))
            raise err
        if not zbox.get("range_done"):
            raise err                                          # the collective itself failed: nothing more to take part in
    # everything below is needed by rank 0 only (it assembles the stream): gathers, not all-gathers; the one value every
    # rank needs is the size of the largest block of strings (the gather's common buffer size) — and the status word
    st = ex.all_reduce_min("all_reduce y bytes", torch.tensor([-int(rec[:, 0].sum()), 0 if err is None else -1], dtype=torch.int32)).cpu()
    if err is not None:
        raise err
    if int(st[1]) < 0:
        if "thread" in zbox:
            zbox["thread"].join()
        raise PeerFailure("a peer rank failed after the z leg of compress_hyper_sharded (its y strings do not exist)")
    cap = -int(st[0])
    cap = max(16, -(-cap // 16) * 16)
    rec_all = ex.gather("gather per-cube records", rec)
    s_all = ex.gather("gather y strings", _pad_to(ex.put(_bytes_tensor(y_strings)), cap))
    if rank != 0:
        return None
    rec_all = rec_all.cpu().numpy().reshape(world, bmax, 4)
    s_all = s_all.cpu().numpy()
    ys, rows = [], []
    for r in range(world):
        rlo, rhi = shard_range(B, r, world)
        off = 0
        for i in range(rhi - rlo):
            n = int(rec_all[r, i, 0])
            ys.append(s_all[r, off:off + n].tobytes())
            off += n
        rows.append(rec_all[r, :rhi - rlo])
    rows = np.concatenate(rows)
    zbox["thread"].join()
    if "error" in zbox:
        raise zbox["error"]
    z_string, z_min, z_max = zbox["coded"]
    out = (ys, rows[:, 1].astype(np.int32), rows[:, 2].astype(np.int32), np.array((1,) + tuple(y_tail), np.int32), z_string,
           z_min, z_max, np.array(zbox["shape"], np.int32))
    if points_numbers is not None:
        out += (rows[:, 3].astype(np.uint16),)
    return out


def _pack_bits(mask):
    """uint8 0/1 tensor (any shape, size % 8 == 0) -> uint8 bytes, MSB first like numpy.packbits"""
    m = mask.reshape(-1, 8).to(torch.uint8)
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device=m.device)
    return (m * w).sum(dim=1, dtype=torch.int32).to(torch.uint8)


def decompress_hyper_sharded(stream, ops, points_numbers=None, rho=1.0, group=None, exchange=None, packed=False,
                             gather_all=False):
    """`stream` = the tuple compress_hyper returns (only rank 0's copy is read).  Returns on rank 0 either the
    logits of all cubes [B,cs,cs,cs,1] (points_numbers is None) or the uint8 occupancy masks after the
    per-cube top-k (1 bit per voxel on the wire: 32 KiB per 64^3 cube); None on the other ranks.  packed=True leaves
    the gathered masks as they travelled: (uint8 tensor [B, vox/8] on the collective device, cube shape).
    Only rank 0 merges the cubes and writes the ply (test.py), so the results are GATHERED to rank 0 (each rank sends its
    block once: (W-1)/W of the masks arrive at rank 0); gather_all=True all-gathers them instead and EVERY rank returns
    the whole cloud (W times the traffic).  A rank whose local decode fails reports it in a status all_reduce before the
    gather: it re-raises its error, the others raise PeerFailure, none waits."""
    collect = (lambda name, t: ex.all_gather("all_" + name, t)) if gather_all else (lambda name, t: ex.gather(name, t))
    ex = exchange or Exchange(group)
    rank, world = ex.rank, ex.world
    head = torch.zeros(16, dtype=torch.int64)
    z_cat = rec = s_cat = None
    err = None
    if rank == 0:
        try:
            y_strings, y_min_vs, y_max_vs, y_shape, z_string, z_min_v, z_max_v, z_shape = stream[:8]
            B = len(y_strings)
            rec = np.zeros((B, 4), np.int32)
            rec[:, 0] = [len(s) for s in y_strings]
            rec[:, 1], rec[:, 2] = np.asarray(y_min_vs), np.asarray(y_max_vs)
            if points_numbers is not None:
                rec[:, 3] = np.asarray(points_numbers)
            s_cat = _bytes_tensor(y_strings)
            z_cat = _bytes_tensor([bytes(z_string)])
            head[0], head[1], head[2] = B, s_cat.numel(), int(points_numbers is not None)
            head[3:8] = torch.as_tensor(np.asarray(y_shape, np.int64))
            head[8:13] = torch.as_tensor(np.asarray(z_shape, np.int64))
            head[13], head[14], head[15] = z_cat.numel(), int(z_min_v), int(z_max_v)
        except BaseException as e:                             # noqa: BLE001 — a malformed stream: tell the peers, then raise
            err, head = e, torch.full((16,), -1, dtype=torch.int64)
    head = ex.broadcast("broadcast header", head).cpu().numpy()
    if err is not None:
        raise err
    if int(head[0]) < 0:
        raise PeerFailure("rank 0 could not read the stream handed to decompress_hyper_sharded")
    B, total, have_nums = int(head[0]), int(head[1]), bool(head[2])
    y_shape, z_shape = head[3:8].astype(np.int32), head[8:13]
    if rank != 0:
        z_cat = torch.empty(int(head[13]), dtype=torch.uint8)
        rec = np.zeros((B, 4), np.int32)
        s_cat = torch.empty(total, dtype=torch.uint8)
    z_cat = ex.broadcast("broadcast z string", z_cat).cpu().numpy()
    rec = ex.broadcast("broadcast per-cube records", rec).cpu().numpy().reshape(B, 4)
    s_cat = ex.broadcast("broadcast y strings", s_cat).cpu().numpy()
    ranges = decode_ranges(B, world)
    lo, hi = ranges[rank]
    nb, bmax = hi - lo, max(h - l for l, h in ranges)
    side = 4 * int(y_shape[1])
    cube_shape = (side, side, side, 1)
    masks = None
    try:
        # the z symbols of cubes [0, hi): this rank's prefix of the one sequential stream
        if hi > lo:
            z_pre = ops.decode_z(z_cat.tobytes(), int(head[14]), int(head[15]), np.concatenate([[hi], z_shape[1:]]).astype(np.int32))
            z_pre = z_pre if torch.is_tensor(z_pre) else torch.from_numpy(np.asarray(z_pre))
            z_loc = z_pre.reshape(hi, *[int(v) for v in z_shape[1:]])[lo:hi]
        else:
            z_loc = torch.zeros((0,) + tuple(int(v) for v in z_shape[1:]))
        offs = np.concatenate([[0], np.cumsum(rec[:, 0].astype(np.int64))])
        strings = [s_cat[offs[i]:offs[i + 1]].tobytes() for i in range(lo, hi)]
        raw = ops.decode_local(z_loc, strings, rec[lo:hi, 1], rec[lo:hi, 2], y_shape)      # tensor (HipOps) or ndarray
        logits = raw if torch.is_tensor(raw) else torch.from_numpy(np.asarray(raw))
        if hi > lo:                                # (a rank without cubes still takes part in the gather)
            cube_shape = tuple(int(v) for v in logits.shape[1:])
        if have_nums:
            masks = ops.classify(raw, rec[lo:hi, 3], rho) if nb else np.zeros((0,) + cube_shape, np.uint8)
    except BaseException as e:                                 # noqa: BLE001 — re-raised once the peers know
        err = e
    # a rank whose local decode failed says so BEFORE the gather its peers would otherwise wait in (8 bytes, one latency)
    st = ex.all_reduce_min("all_reduce decode status", torch.tensor([0 if err is None else -1], dtype=torch.int32)).cpu()
    if err is not None:
        raise err
    if int(st[0]) < 0:
        raise PeerFailure("a peer rank failed in its block of decompress_hyper_sharded")
    vox = int(np.prod(cube_shape))
    if not have_nums:
        out = collect("gather logits", _pad_to(ex.put(logits.to(torch.float32)), bmax * vox))
        if out is None:
            return None
        out = out.cpu().numpy().reshape(world, bmax, vox)
        parts = [out[r, :ranges[r][1] - ranges[r][0]] for r in range(world)]
        return np.concatenate(parts).reshape((B,) + cube_shape)
    assert vox % 8 == 0
    masks = masks if torch.is_tensor(masks) else torch.from_numpy(np.asarray(masks, np.uint8))
    bits = ex.put(_pack_bits(masks)) if nb else torch.zeros(0, dtype=torch.uint8, device=ex.device)   # packed where the masks live
    out = collect("gather occupancy bit masks", _pad_to(bits, bmax * vox // 8))
    if out is None:                                # gathered to rank 0 only
        return None
    out = out.reshape(world, bmax, vox // 8)
    parts = [out[r, :ranges[r][1] - ranges[r][0]] for r in range(world)]
    if packed:
        return torch.cat(parts), cube_shape
    return np.unpackbits(torch.cat(parts).cpu().numpy(), axis=1).reshape((B,) + cube_shape)
