"""Host-side timeline of one pipelined compress_hyper + decompress_hyper step (the bench workload): wall-clock
intervals of the host calls that gate the GPU (range coder batches, event waits, D2H syncs), per pipeline thread.
    python tools/host_timeline.py [min_us]"""
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

LOG = []
T0 = [0.0]


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, label))
    setattr(obj, name, g)


class HostProxy(object):
    """times every C call on the host library"""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = getattr(self._lib, name)

        def g(*a):
            t = time.perf_counter()
            try:
                return f(*a)
            finally:
                LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, "host." + name))
        return g


def main(min_us=50.0, profile="sparse"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    checkpoint._CACHE["bench"] = (checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00")) if profile == "trained"
                                  else synthetic.make_weights(seed=1300, profile=profile))
    pts = synthetic.make_cloud(seed=1300)
    cubes, _, _ = process.preprocess_points(pts, 1.0, 64, 64)

    def step():
        out = transform.compress_hyper(cubes, model, "bench")
        return transform.decompress_hyper(*out, model, "bench")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    host = _lib.host()
    proxy = HostProxy(host)
    _lib.host = lambda: proxy
    wrap(torch.cuda.Event, "synchronize", "event.synchronize")
    wrap(torch.Tensor, "cpu", "tensor.cpu")
    wrap(cem.SymmetricConditional, "compress_cubes")
    wrap(em.EntropyBottleneck, "compress_async")
    wrap(em.EntropyBottleneck, "decompress_async")
    wrap(transform, "_compress_hyper_pipes")
    wrap(transform, "_run_pipes")
    c = transform.get_codec(model, "bench")
    for n in ("analysis_transform", "synthesis_transform", "hyper_encoder", "hyper_decoder"):
        wrap(getattr(c, n).__class__, "__call__", n) if False else None
    T0[0] = time.perf_counter()
    step()
    torch.cuda.synchronize()
    end = time.perf_counter() - T0[0]
    for a, b, th, lab in sorted(LOG):
        if (b - a) * 1e6 >= min_us:
            print("%8.2f %8.2f  %7.2f ms  %-14s %s" % (a * 1e3, b * 1e3, (b - a) * 1e3, th[-14:], lab))
    print("step %.2f ms" % (end * 1e3))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 50.0, sys.argv[2] if len(sys.argv) > 2 else "sparse")
