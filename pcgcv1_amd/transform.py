"""Host orchestration of the learned codec — same entry points as the reference's
transform.py:

  compress_hyper(cubes, model, ckpt_dir, decompress=False)                 transform.py:91-197
  decompress_hyper(y_strings, y_min_vs, y_max_vs, y_shape, z_strings,
                   z_min_v, z_max_v, z_shape, model, ckpt_dir)              transform.py:200-259
  compress_factorized / decompress_factorized                              transform.py:24-87

`model` is a module exposing AnalysisTransform / SynthesisTransform / HyperEncoder /
HyperDecoder (pcgcv1_amd.models.model_voxception), exactly how the reference passes the module
chosen by importlib (test.py:72).  Differences that are deliberate: cubes are processed as a
batch on the GPU instead of tf.map_fn(parallel_iterations=1); the per-cube range coding runs on a
host thread pool; stage timers keep the reference's stage names (printed with `verbose=True`).
Return values are numpy arrays / bytes (the reference returns eager tensors the caller .numpy()s).
"""
import os
import threading
import time

import numpy as np
import torch

from . import _lib, checkpoint
from .models.conditional_entropy_model import SymmetricConditional, decode_slices
from .models.entropy_model import EntropyBottleneck

LOWER_BOUND = 1e-9          # transform.py:145, 232
# Cubes are independent (one cube per call in the reference), so a large batch runs as PCGC_PIPES contiguous groups,
# each on its own HIP stream driven by its own host thread: while one group's strings are range-coded on the host
# the other group's kernels keep the GPU busy.  The bitstream is unchanged (per-cube y strings in cube order, ONE z
# string over all cubes).  Measured on the 205-cube batch: 1 pipe 72 ms, 2 pipes see DESIGN.md §6.
_PIPES = int(os.environ.get("PCGC_PIPES", "2"))
_EARLY_RANGES = int(os.environ.get("PCGC_EARLY_RANGES", "1"))   # experiment knobs (measured: DESIGN.md §9)
_si = os.environ.get("PCGC_SWITCH_INTERVAL_US")
if _si:
    import sys
    sys.setswitchinterval(int(_si) * 1e-6)
_MIN_GROUP = 48             # cubes per pipeline below which splitting costs more than it hides
_TAIL_SLICE = int(os.environ.get("PCGC_TAIL_SLICE", "24"))      # last decoder slice of a pipeline when the caller streams the tail


def _groups(B, n=None):
    n = max(1, min(_PIPES if n is None else n, B // _MIN_GROUP))
    base, rem = divmod(B, n)
    out, lo = [], 0
    for i in range(n):
        hi = lo + base + (1 if i < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


_DEC_INTERLEAVE = os.environ.get("PCGC_DEC_INTERLEAVE", "1") != "0"


def _decode_plan(B, groups, tail=0, interleave=True):
    """Per pipeline, the slices (ranges of the batch) it decodes, in order.  PCGC_DEC_INTERLEAVE=0: each pipeline its contiguous
    group, cut by decode_slices (a short first slice, then the rest).  Default: the pipelines' short first slices are the first
    cubes of the cloud in z order, their long slices share the rest as the groups would."""
    n = len(groups)
    per = [decode_slices(hi - lo, tail=tail) for lo, hi in groups]
    if not (_DEC_INTERLEAVE and interleave) or n < 2 or any(len(p) < 2 for p in per):
        return [[(lo + a, lo + b) for a, b in p] for (lo, hi), p in zip(groups, per)]
    firsts = [p[0][1] - p[0][0] for p in per]
    plan, at = [], 0
    for f in firsts:                                         # the first slices, one per pipeline, from the start of the cloud
        plan.append([(at, at + f)])
        at += f
    for i, p in enumerate(per):                              # the rest: each pipeline's later slices keep their lengths
        for a, b in p[1:]:
            plan[i].append((at, at + b - a))
            at += b - a
    assert at == B
    return plan


def _pipe_streams(codec, n):
    """n side streams that belong to the calling thread's current stream (process-wide, _lib.side_stream): two calls running
    at once on two streams (compress_hyper_ahead) do not queue behind each other"""
    cur = torch.cuda.current_stream()
    return [_lib.side_stream("pipe%d" % i, cur) for i in range(n)]


def _run_pipes(codec, groups, fn):
    """fn(i, lo, hi) for every group, each on its own thread + stream; the streams start after everything already
    queued on the caller's stream and the caller's stream continues after all of them.  Re-raises the first error."""
    main = torch.cuda.current_stream()
    streams = _pipe_streams(codec, len(groups))
    ready = torch.cuda.Event()
    ready.record(main)
    done = [torch.cuda.Event() for _ in groups]
    errs = []

    dev = main.device                                  # the caller's device: worker threads start on device 0

    def body(i, lo, hi):
        try:
            _lib.bind_device(dev)
            with torch.cuda.stream(streams[i]):
                streams[i].wait_event(ready)
                fn(i, lo, hi)
                done[i].record()
        except BaseException as e:                     # noqa: BLE001 — re-raised on the caller's thread
            errs.append(e)
    # persistent workers (a fresh thread per pipeline and call costs 0.2-0.4 ms before its first launch).  Every pipeline of
    # a call must run at once (they meet at a barrier): the "pipe" pool holds nothing but pipeline bodies (the jobs they
    # submit and wait for run on the "job" pool), and a call that would not find all its workers free takes dedicated
    # threads instead of queueing half of its pipelines behind other calls'
    with _LOCK:
        _INFLIGHT[0] += len(groups)
        dedicated = _INFLIGHT[0] > _PIPE_WORKERS
    try:
        if dedicated:
            ths = [threading.Thread(target=body, args=(i, lo, hi), name="pcgc-pipe-extra") for i, (lo, hi) in enumerate(groups)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        else:
            for f in [_workers().submit(body, i, lo, hi) for i, (lo, hi) in enumerate(groups)]:
                f.result()
    finally:
        with _LOCK:
            _INFLIGHT[0] -= len(groups)
    if errs:          # a sibling's BrokenBarrierError is a consequence, not the cause
        real = [e for e in errs if not isinstance(e, threading.BrokenBarrierError)]
        raise (real or errs)[0]
    for e in done:
        main.wait_event(e)


def _workers():
    return _lib.workers("pipe")


_CODECS = {}
_LOCK = threading.Lock()
_INFLIGHT = [0]             # pipeline bodies of all running _run_pipes calls
_PIPE_WORKERS = 64          # size of _lib.workers("pipe")


class Codec(object):
    """The five operators bound to one checkpoint (built once per (model, ckpt_dir))."""

    def __init__(self, model, ckpt_dir):
        w = checkpoint.load(ckpt_dir)
        # the device the operators' weights live on (one process per GPU: the rank's device); every thread that launches
        # for this codec binds it first (_run_pipes, the z leg, process.StreamedPostprocess)
        self.device = _lib.require_gpu()
        self.analysis_transform = model.AnalysisTransform().load_weights(w)
        self.synthesis_transform = model.SynthesisTransform().load_weights(w)
        # hyperprior parts: absent from factorized checkpoints (transform.py:35-38) and from models.model_simple
        self.hyper_encoder = self.hyper_decoder = self.entropy_bottleneck = None
        if hasattr(model, "HyperEncoder") and "hyper_encoder/conv1/kernel" in w:
            self.hyper_encoder = model.HyperEncoder().load_weights(w)
            self.hyper_decoder = model.HyperDecoder().load_weights(w)
            self.entropy_bottleneck = EntropyBottleneck().load_weights(w, "estimator")
        self.conditional_entropy_model = SymmetricConditional()
        self.timers = {}
        self.last_path = {}                     # which branch the last compress / decompress call took

    def require_hyper(self):
        if self.hyper_encoder is None:
            raise ValueError("this model / checkpoint has no hyperprior (hyper_encoder, hyper_decoder, 8-channel estimator): "
                             "use --mode=factorized")
        return self


def get_codec(model, ckpt_dir):
    key = (getattr(model, "__name__", str(model)), str(ckpt_dir))
    with _LOCK:
        if key not in _CODECS:
            _CODECS[key] = Codec(model, ckpt_dir)
        return _CODECS[key]


class _Stage(object):
    def __init__(self, timers, name, verbose):
        self.t, self.name, self.verbose = timers, name, verbose

    def __enter__(self):
        torch.cuda.synchronize()
        self.t0 = time.time()

    def __exit__(self, *a):
        torch.cuda.synchronize()
        self.t[self.name] = time.time() - self.t0
        if self.verbose:
            print("{}: {}s".format(self.name, round(self.t[self.name], 4)))


def _to_device(cubes):
    dev = _lib.require_gpu()
    if torch.is_tensor(cubes):
        return cubes.to(dev, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(cubes, np.float32)).to(dev)


def _compress_hyper_pipes(c, x, groups, code_z=True, z_hook=None):
    """code_z=False (sharded encoder): the single z string is coded elsewhere, over the cubes of all ranks; the rounded
    hyper-latents come back instead of the string.  z_hook(z_hat of the whole block) is then called by one pipeline as soon
    as every pipeline's hyper-latents exist — the exchange and the coding of the z string run while the y strings of the
    block are still being produced."""
    n = len(groups)
    zev, res, zbox = [None] * n, [None] * n, {}
    barrier = threading.Barrier(n)
    zstream = _pipe_streams(c, n + 1)[n]
    # every pipeline's hyper encoder writes its cubes' z into ONE pre-sized buffer (and the rounded values likewise): the
    # single z stream is coded from it without a concatenation pass
    cs = int(x.shape[1])
    B_all = groups[-1][1]
    z_all = torch.empty((B_all, cs // 8, cs // 8, cs // 8, 8), dtype=torch.float32, device=x.device)
    zh_all = torch.empty_like(z_all)

    def z_work():
        _lib.bind_device(c.device)
        zdone = torch.cuda.Event()
        with torch.cuda.stream(zstream):
            for e in zev:
                zstream.wait_event(e)
            if code_z:
                zbox["job"] = c.entropy_bottleneck.compress_async(z_all)
            else:
                z_hook(zh_all)
            zdone.record()
        zbox["done"] = zdone

    def work(i, lo, hi):
        try:
            _lib.mark("enc pipe %d start" % i)
            ys = c.analysis_transform(x[lo:hi])
            _lib.mark("enc pipe %d analysis queued" % i)
            # rounding + per-cube symbol ranges are queued NOW and travel to the host under the hyper encoder / decoder
            # launches: compress_cubes finds them there instead of stalling on a round trip of its own
            ranges = c.conditional_entropy_model.start_ranges(ys) if _EARLY_RANGES else None
            zs = c.hyper_encoder(ys, out=z_all[lo:hi])
            z_hats = c.entropy_bottleneck.quantize_into(zs, zh_all[lo:hi])      # round-half-even: what __call__(training=False) returns as values
            zev[i] = torch.cuda.Event()
            zev[i].record()
            _lib.mark("enc pipe %d hyper encoder queued" % i)
            if (code_z or z_hook) and barrier.wait() == 0:    # one pipeline starts the single z stream as soon as every z exists
                # ... before its own hyper decoder (the z string is the longest serial piece of the tail) and on a stream of
                # its own: the round trip of the z symbols waits for the hyper encoders only.  The pipeline thread that does it is
                # held until the symbols are on the host (= until every pipeline's analysis is through) and queues its own hyper
                # decoder ~12 ms late; handing the z leg to a worker thread instead was measured twice — round 4: no difference;
                # round 5 (tools/exp/t_slow_steps.py shows the lag): compress_hyper 16.05 -> 16.48 ms, four interleaved pairs
                z_work()
            _lib.mark("enc pipe %d past the z barrier" % i)
            locs, scales = c.hyper_decoder(z_hats, lower_bound=LOWER_BOUND)
            res[i] = (c.conditional_entropy_model.compress_cubes(ys, locs, scales, ranges=ranges)
                      + (tuple(ys.shape[1:]), tuple(zs.shape[1:])))
            _lib.mark("enc pipe %d strings done" % i)
        except BaseException:
            barrier.abort()
            raise
    _run_pipes(c, groups, work)
    if "done" in zbox:
        torch.cuda.current_stream().wait_event(zbox["done"])
    y_strings = [s_ for r in res for s_ in r[0]]
    y_min_vs = np.concatenate([r[1] for r in res]).astype(np.int32)
    y_max_vs = np.concatenate([r[2] for r in res]).astype(np.int32)
    if not code_z:
        return zh_all, y_strings, y_min_vs, y_max_vs, res[0][3]
    _lib.mark("enc pipes joined")
    z_strings, z_min_v, z_max_v = zbox["job"]()
    _lib.mark("enc z string done")
    B = groups[-1][1]
    return (y_strings, y_min_vs, y_max_vs, np.array((1,) + res[0][3], np.int32), z_strings, z_min_v, z_max_v,
            np.array((B,) + res[0][4], np.int32))


def compress_block(c, cubes, z_hook=None):
    """One rank's share of a sharded encode (sharding.HipOps): everything of compress_hyper except the z string.
    -> (z_hat float tensor [b,...] on the device, y_strings, y_min_vs, y_max_vs, shape of one cube's y); the same host
    pipelines as compress_hyper when the block is large enough.  z_hook(z_hat): called once, as early as the block's
    hyper-latents exist (before the y strings are coded)."""
    x = _to_device(cubes)
    if int(x.shape[0]) == 0:                                  # a rank without cubes (fewer cubes than ranks)
        cs = int(x.shape[1])
        z0 = torch.zeros((0, cs // 8, cs // 8, cs // 8, 8), device=x.device)
        if z_hook:
            z_hook(z0)
        return (z0, [], np.zeros(0, np.int32), np.zeros(0, np.int32), (cs // 4, cs // 4, cs // 4, 16))
    groups = _groups(int(x.shape[0]))
    c.last_path = {"call": "compress_block", "cubes": int(x.shape[0]), "pipelines": len(groups)}
    if len(groups) > 1:
        return _compress_hyper_pipes(c, x, groups, code_z=False, z_hook=z_hook)
    ys = c.analysis_transform(x)
    zs = c.hyper_encoder(ys)
    z_hats, _ = c.entropy_bottleneck(zs, False)
    if z_hook:
        z_hook(z_hats)
    locs, scales = c.hyper_decoder(z_hats, lower_bound=LOWER_BOUND)
    y_strings, y_min_vs, y_max_vs = c.conditional_entropy_model.compress_cubes(ys, locs, scales)
    return z_hats, y_strings, y_min_vs, y_max_vs, tuple(ys.shape[1:])


def decompress_block(c, z_hat, y_strings, y_min_vs, y_max_vs, y_shape):
    """One rank's share of a sharded decode: hyper decoder -> range decoding -> synthesis for cubes whose z-hat is
    already known (the rank decoded its prefix of the z string itself, sharding.decompress_hyper_sharded).  -> logits [b,cs,cs,cs,1] on the device."""
    dev = _lib.require_gpu()
    z = (z_hat if torch.is_tensor(z_hat) else torch.from_numpy(np.asarray(z_hat))).to(dev, torch.float32).contiguous()
    y_strings = list(y_strings)
    y_min_vs, y_max_vs = np.asarray(y_min_vs), np.asarray(y_max_vs)
    groups = _groups(len(y_strings))
    row_bytes = 2 * (int((y_max_vs - y_min_vs).max()) + 1) if len(y_strings) else 0      # one CDF row on its way to the host
    c.last_path = {"call": "decompress_block", "cubes": len(y_strings), "pipelines": len(groups)}
    side = 4 * int(y_shape[1])
    xs = torch.empty((len(y_strings), side, side, side, 1), dtype=torch.float32, device=dev)

    def work(i, lo, hi):
        def hd(a, b):                                         # per entropy slice (results do not depend on the batch cut)
            return c.hyper_decoder(z[lo + a:lo + b].contiguous(), lower_bound=LOWER_BOUND)
        for a, b, y in c.conditional_entropy_model.decompress_slices(y_strings[lo:hi], hd, None, y_min_vs[lo:hi],
                                                                     y_max_vs[lo:hi], y_shape, slices=decode_slices(hi - lo, row_bytes=row_bytes)):
            c.synthesis_transform(y, out=xs[lo + a:lo + b])        # straight into the batch
    if len(groups) > 1:
        _run_pipes(c, groups, work)
    elif len(y_strings):
        work(0, 0, len(y_strings))
    return xs


def compress_hyper(cubes, model, ckpt_dir, decompress=False, verbose=False, profile_stages=False):
    c = get_codec(model, ckpt_dir).require_hyper()
    t = c.timers
    stage = (lambda n: _Stage(t, n, verbose)) if (verbose or profile_stages) else (lambda n: _Null())
    x = _to_device(cubes)
    groups = _groups(int(x.shape[0]))
    c.last_path = {"call": "compress_hyper", "cubes": int(x.shape[0]), "pipelines": 1}
    if len(groups) > 1 and not (decompress or verbose or profile_stages):
        c.last_path["pipelines"] = len(groups)
        return _compress_hyper_pipes(c, x, groups)
    with stage("Analysis Transform"):
        ys = c.analysis_transform(x)
    with stage("Hyper Encoder"):
        zs = c.hyper_encoder(ys)
    z_hats, _ = c.entropy_bottleneck(zs, False)                       # transform.py:134
    with stage("Hyper Decoder"):
        locs, scales = c.hyper_decoder(z_hats, lower_bound=LOWER_BOUND)
    with stage("Entropy Encode (Hyper)"):
        # the single z string (entropy_model.py:249-259) is sequential host work: code it on a helper thread
        # while the device builds the y CDFs and the pool codes the y strings
        z_job = c.entropy_bottleneck.compress_async(zs)
        z_shape = np.array(zs.shape, np.int32)
    with stage("Entropy Encode"):
        y_strings, y_min_vs, y_max_vs = c.conditional_entropy_model.compress_cubes(ys, locs, scales)
        y_shape = np.array((1,) + tuple(ys.shape[1:]), np.int32)
        z_strings, z_min_v, z_max_v = z_job()
    out = (y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape)
    if decompress:
        with stage("Entropy Decode"):
            y_dec = c.conditional_entropy_model.decompress_cubes(y_strings, locs, scales, y_min_vs, y_max_vs, y_shape)
        with stage("Synthesis Transform"):
            x_dec = c.synthesis_transform(y_dec)
        return out + (x_dec,)
    return out


def decompress_hyper(y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape, model, ckpt_dir,
                     verbose=False, profile_stages=False, on_slice=None):
    """on_slice(lo, hi, logits of cubes lo..hi): called for every entropy slice right after its synthesis has been QUEUED,
    on the thread and stream that queued it (record an event there, do not wait: process.StreamedPostprocess).  Slices
    arrive out of order (one sequence per host pipeline)."""
    c = get_codec(model, ckpt_dir).require_hyper()
    t = c.timers
    stage = (lambda n: _Stage(t, n, verbose)) if (verbose or profile_stages) else (lambda n: _Null())
    y_strings = list(y_strings)
    groups = _groups(len(y_strings))
    c.last_path = {"call": "decompress_hyper", "cubes": len(y_strings), "pipelines": 1}
    if len(groups) > 1 and not (verbose or profile_stages):
        c.last_path["pipelines"] = len(groups)
        # the z stream is sequential: a helper thread decodes it and each pipeline starts as soon as its cubes' symbols
        # are final (the first group after 1 / n of the decoding time)
        z_part = c.entropy_bottleneck.decompress_async(z_strings, z_min_v, z_max_v, z_shape, int(z_shape[-1]))
        y_min_vs, y_max_vs = np.asarray(y_min_vs), np.asarray(y_max_vs)
        row_bytes = 2 * (int((y_max_vs - y_min_vs).max()) + 1)                              # one CDF row on its way to the host
        side = 4 * int(y_shape[1])
        xs = torch.empty((len(y_strings), side, side, side, 1), dtype=torch.float32, device=_lib.require_gpu())

        # Which cubes a pipeline decodes.  Every pipeline starts with a short slice and nothing of it runs before ITS cubes' share
        # of the one sequential z stream is decoded (18 us per cube): with contiguous halves the second pipeline's first slice sat
        # behind the z symbols of 127 cubes (2.2 ms).  The short first slices are therefore the FIRST cubes of the cloud, one per
        # pipeline in z order, and the pipelines' long slices follow — a pipeline's cubes are not contiguous, so its strings are
        # gathered into a list of their own and slice boundaries map back to positions in the batch.
        # (a caller that streams the tail — the CLI — keeps contiguous groups: 29.7 ms per decompress against 29.9, its last slices'
        # tails are what it waits for.)
        plan = _decode_plan(len(y_strings), groups, tail=_TAIL_SLICE if on_slice else 0, interleave=on_slice is None)

        def work(i, lo_, hi_):
            _lib.mark("dec pipe %d start" % i)
            mine = plan[i]                                           # [(g_lo, g_hi)] in the order this pipeline decodes them
            idx = np.concatenate([np.arange(a, b) for a, b in mine])
            local, at = [], 0
            for a, b in mine:
                local.append((at, at + b - a))
                at += b - a
            where = dict(zip(local, mine))

            # hyper decoder per entropy slice: a slice waits only for the z symbols up to ITS last cube
            def hd(a, b):
                ga, gb = where[(a, b)]
                z = z_part(ga, gb)
                _lib.mark("dec pipe %d z[%d:%d] on device" % (i, ga, gb))
                return c.hyper_decoder(z, lower_bound=LOWER_BOUND)

            for a, b, y in c.conditional_entropy_model.decompress_slices([y_strings[k] for k in idx], hd, None, y_min_vs[idx],
                                                                         y_max_vs[idx], y_shape, slices=local):
                ga, gb = where[(a, b)]
                c.synthesis_transform(y, out=xs[ga:gb])              # straight into the batch
                if on_slice:
                    on_slice(ga, gb, xs[ga:gb])
        _run_pipes(c, groups, work)
        return xs
    with stage("Entropy Decoder (Hyper)"):
        zs = c.entropy_bottleneck.decompress(z_strings, z_min_v, z_max_v, z_shape, int(z_shape[-1]))
    with stage("Hyper Decoder"):
        locs, scales = c.hyper_decoder(zs, lower_bound=LOWER_BOUND)
    with stage("Entropy Decoder + Synthesis Transform"):
        # slice pipeline: the host range-decodes slice k+1 while the device synthesises slice k
        xs = None
        for lo, hi, y in c.conditional_entropy_model.decompress_slices(y_strings, locs, scales, y_min_vs, y_max_vs,
                                                                        y_shape):
            x = c.synthesis_transform(y)
            if xs is None:
                xs = torch.empty((len(y_strings),) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
            xs[lo:hi] = x
            if on_slice:
                on_slice(lo, hi, xs[lo:hi])
        if xs is None:
            xs = c.synthesis_transform(torch.empty((0,) + tuple(int(v) for v in y_shape[1:]), device=locs.device))
    return xs


class _Ahead(object):
    """compress_hyper on a helper thread with its own HIP stream (and, through _pipe_streams, its own pipeline streams)."""

    def __init__(self, cubes, model, ckpt_dir):
        self._box = {}
        x = _to_device(cubes)
        ready = torch.cuda.Event()
        ready.record()                                        # whatever produced the cubes on the caller's stream
        dev = torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(_side_stream()):
                    torch.cuda.current_stream().wait_event(ready)
                    self._box["out"] = compress_hyper(x, model, ckpt_dir)
            except BaseException as e:                       # noqa: BLE001 — re-raised by result()
                self._box["err"] = e
        self._thread = threading.Thread(target=work, name="compress-ahead")
        self._thread.start()

    def result(self):
        self._thread.join()
        if "err" in self._box:
            raise self._box["err"]
        return self._box["out"]


_SIDE = []


def _side_stream():
    with _LOCK:
        if not _SIDE:
            _SIDE.append(torch.cuda.Stream())
        return _SIDE[0]


def compress_hyper_ahead(cubes, model, ckpt_dir):
    """compress_hyper(cubes, model, ckpt_dir) started now, on its own thread and streams; `.result()` joins it and returns
    the same tuple.  For jobs over several clouds or rate points (eval.py's loop, a directory of frames): at the end of an
    encode and the start of a decode the GPU waits for the host (range coding, the sequential z string), and the next
    cloud's analysis fills that gap.  The bytes do not depend on what else runs (every kernel's summation order is fixed)."""
    return _Ahead(cubes, model, ckpt_dir)


def roundtrip_stream(batches, model, ckpt_dir):
    """for cubes in batches: yield (compress_hyper(cubes), decompress_hyper(of those streams)) with the encode of the next
    batch running while the current one decodes.  `batches`: an iterable of cube tensors / arrays (it is advanced one
    batch ahead of what has been yielded)."""
    it = iter(batches)
    try:
        ahead = compress_hyper_ahead(next(it), model, ckpt_dir)
    except StopIteration:
        return
    while ahead is not None:
        out = ahead.result()
        try:
            ahead = compress_hyper_ahead(next(it), model, ckpt_dir)
        except StopIteration:
            ahead = None
        try:
            xs = decompress_hyper(*out, model, ckpt_dir)
        except BaseException:
            if ahead is not None:
                try:
                    ahead.result()                            # never leave a running encode behind an exception
                except BaseException:                         # noqa: BLE001
                    pass
            raise
        yield out, xs


def compress_factorized(cubes, model, ckpt_dir, verbose=False):
    """transform.py:24-56."""
    c = get_codec(model, ckpt_dir)
    ys = c.analysis_transform(_to_device(cubes))
    strings, min_v, max_v = c.entropy_bottleneck_y(ckpt_dir, int(ys.shape[-1])).compress(ys)
    return strings, min_v, max_v, np.array(ys.shape, np.int32)


def decompress_factorized(strings, min_v, max_v, shape, model, ckpt_dir, verbose=False):
    """transform.py:58-87."""
    c = get_codec(model, ckpt_dir)
    ys = c.entropy_bottleneck_y(ckpt_dir, int(shape[-1])).decompress(strings, min_v, max_v, shape, int(shape[-1]))
    return c.synthesis_transform(ys)


def _entropy_bottleneck_y(self, ckpt_dir, channels=None):
    """Factorized mode codes the latents y with an EntropyBottleneck stored under the 'estimator' key of a
    *factorized* checkpoint (transform.py:35-38): 16 channels for model_voxception, 32 for model_simple.  When the
    checkpoint's estimator has another width (a hyper checkpoint: 8 channels for z) a bottleneck with the
    reference's initialisers is built instead — for seeded synthetic weights only; a real checkpoint without a matching
    estimator is an error (the reference's restore would fail on it too), never a silently untrained prior."""
    eb = getattr(self, "_eb_y", None)
    if eb is None:
        w = checkpoint.load(ckpt_dir)
        eb = EntropyBottleneck()
        have = int(w["estimator/matrix_0"].shape[0]) if "estimator/matrix_0" in w else None
        if have is not None and (channels is None or have == channels) and have != 8:
            eb.load_weights(w, "estimator")
        elif str(ckpt_dir) == "" or str(ckpt_dir).startswith("synthetic") or str(ckpt_dir) in checkpoint._CACHE_SYNTHETIC:
            eb.build(channels or 16, rng=np.random.default_rng(1300))
        else:
            raise ValueError("--mode=factorized needs a factorized checkpoint: %r holds %s, the latents have %s channels"
                             % (str(ckpt_dir), "no estimator" if have is None else "a %d-channel estimator" % have, channels))
        self._eb_y = eb
    return eb


Codec.entropy_bottleneck_y = _entropy_bottleneck_y


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
