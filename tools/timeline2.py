"""Host AND device timeline of one pipelined compress_hyper + decompress_hyper step (the bench workload).
H rows: wall-clock intervals of host calls (range coder batches, event waits, D2H syncs, launches) per thread.
G rows: when the device actually ran what a call queued (HIP events on the call's stream, same clock: the base event is
synchronised at t = 0), per stream.  Shows where the GPU waits for the host at the encode -> decode hand-over.
    python tools/timeline2.py [min_us] [profile] [steps] [copies of the cloud in one call]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcgcv1_amd import _lib, checkpoint, process, synthetic, transform  # noqa: E402
from pcgcv1_amd.models import conditional_entropy_model as cem  # noqa: E402
from pcgcv1_amd.models import entropy_model as em  # noqa: E402
from pcgcv1_amd.models import model_voxception as model  # noqa: E402

LOG, GLOG = [], []
T0 = [0.0]
ON = [False]


def wrap(obj, name, label=None, gpu=False):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        if not ON[0]:
            return f(*a, **k)
        lab = label(a) if callable(label) else label
        if gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, lab))
            if gpu:
                e1.record()
                GLOG.append((e0, e1, int(torch.cuda.current_stream().cuda_stream) & 0xffff, lab))
    setattr(obj, name, g)


class Proxy(object):
    """times every C call on a library; GPU calls also get device intervals"""

    def __init__(self, lib, prefix, gpu):
        self._lib, self._prefix, self._gpu = lib, prefix, gpu
        self._cache = {}

    def __getattr__(self, name):
        if name in self._cache:
            return self._cache[name]
        f = getattr(self._lib, name)
        gpu = self._gpu and name in ("pcgc_laplace_cdf", "pcgc_round_minmax", "pcgc_factorized_likelihood")

        def g(*a):
            if not ON[0]:
                return f(*a)
            if gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t = time.perf_counter()
            try:
                return f(*a)
            finally:
                LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, self._prefix + name))
                if gpu:
                    e1.record()
                    GLOG.append((e0, e1, int(torch.cuda.current_stream().cuda_stream) & 0xffff, self._prefix + name))
        self._cache[name] = g
        return g


def main(min_us=50.0, profile="trained", steps=1, copies=1):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    checkpoint._CACHE["bench"] = (checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00")) if profile == "trained"
                                  else synthetic.make_weights(seed=1300, profile=profile))
    pts = synthetic.make_cloud(seed=1300)
    cubes, _, _ = process.preprocess_points(pts, 1.0, 64, 64)
    if copies > 1:
        cubes = cubes.repeat(copies, 1, 1, 1, 1)             # one large cloud (bench.py's large_cloud figure)

    def step():
        out = transform.compress_hyper(cubes, model, "bench")
        return transform.decompress_hyper(*out, model, "bench")
    host = Proxy(_lib.host(), "host.", False)
    hip = Proxy(_lib.hip(), "hip.", True)
    _lib.host = lambda: host
    _lib.hip = lambda: hip
    _lib._trace = lambda lab: ON[0] and LOG.append((time.perf_counter() - T0[0], time.perf_counter() - T0[0] + 1e-3,
                                                    threading.current_thread().name, "* " + lab))
    wrap(torch.cuda.Event, "synchronize", "event.synchronize")
    wrap(torch.Tensor, "cpu", "tensor.cpu")
    wrap(cem.SymmetricConditional, "compress_cubes")
    wrap(em.EntropyBottleneck, "compress_async")
    wrap(em.EntropyBottleneck, "decompress_async")
    wrap(transform, "_compress_hyper_pipes")
    wrap(transform, "compress_hyper")
    wrap(transform, "decompress_hyper")
    wrap(model._Net, "_forward", lambda a: "%s[%d]" % (a[0].net_name, int(a[1].shape[0])), gpu=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(steps):
        del LOG[:], GLOG[:]
        base = torch.cuda.Event(enable_timing=True)
        base.record()
        base.synchronize()
        T0[0] = time.perf_counter()
        ON[0] = True
        step()
        torch.cuda.synchronize()
        ON[0] = False
        end = time.perf_counter() - T0[0]
        rows = [(a * 1e3, a * 1e3 if lab.startswith("* ") else b * 1e3, "H", th[-12:], lab) for a, b, th, lab in LOG
                if (b - a) * 1e6 >= min_us or lab.startswith("* ")]
        rows += [(base.elapsed_time(e0), base.elapsed_time(e1), "G", "stream %04x" % sid, lab) for e0, e1, sid, lab in GLOG]
        for a, b, kind, who, lab in sorted(rows):
            print("%8.2f %8.2f  %7.2f ms  %s %-12s %s" % (a, b, b - a, kind, who, lab))
        print("step %.2f ms" % (end * 1e3))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 50.0, sys.argv[2] if len(sys.argv) > 2 else "trained",
         int(sys.argv[3]) if len(sys.argv) > 3 else 1, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
