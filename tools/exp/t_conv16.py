"""The stride-1 layers of the training step's 16^3 stage one by one (a batch of 8 cubes: conv_mfma_kernel's 2 x 2-row tiles),
back to back on one stream: us per launch and TFLOP/s per shape, through pcgc_train_conv_fwd on a prepared plan (no packing
inside the timed loop).  GPU box only.
    python tools/exp/t_conv16.py [B] [D]
The reverse pass runs the same kernels on the adjoint shapes, listed here as forward layers of those shapes."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from pcgcv1_amd import _lib  # noqa: E402
from pcgcv1_amd.train_hyper import _TrainLayer  # noqa: E402

SHAPES = [(64, 16, 3), (64, 16, 1), (16, 32, 3), (16, 16, 3), (16, 32, 1), (32, 16, 3), (32, 16, 1), (16, 64, 3), (16, 64, 1)]
PAIRS = [(0, 1), (2, 3), (5, 6)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    lib, dev = _lib.hip(), _lib.require_gpu()
    g = torch.Generator(device="cpu").manual_seed(1)
    ks = [(torch.randn((k, k, k, ci, co), generator=g) * 0.1).to(dev) for ci, co, k in SHAPES]
    gks = [torch.zeros_like(k_) for k_ in ks]
    gbs = [torch.zeros(co, device=dev) for _, co, _ in SHAPES]
    arr = (_TrainLayer * len(SHAPES))()
    for i, (ci, co, k) in enumerate(SHAPES):
        arr[i].kernel, arr[i].dkernel, arr[i].dbias = ks[i].data_ptr(), gks[i].data_ptr(), gbs[i].data_ptr()
        arr[i].Cin, arr[i].Cout, arr[i].ksize, arr[i].stride, arr[i].transposed = ci, co, k, 1, 0
    plan = ctypes.c_void_p()
    _lib.check(lib.pcgc_train_plan_create(ctypes.cast(arr, ctypes.c_void_p), len(SHAPES), ctypes.byref(plan)))
    _lib.check(lib.pcgc_train_plan_prepare(plan, _lib.stream()))
    xs = {c: torch.relu(torch.randn((B, D, D, D, c), generator=g)).to(dev) for c in (16, 32, 64)}
    ys = [torch.empty((B, D, D, D, co), device=dev) for _, co, _ in SHAPES]
    bs = [torch.randn(co, generator=g).to(dev) for _, co, _ in SHAPES]

    def timed(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    total = 0.0
    for i, (ci, co, k) in enumerate(SHAPES):
        def one(i=i, ci=ci):
            _lib.check(lib.pcgc_train_conv_fwd(plan, i, _lib.dptr(xs[ci]), _lib.dptr(bs[i]), _lib.dptr(ys[i]), B, D, 1, _lib.stream()))
        us = timed(one)
        flop = 2.0 * B * D ** 3 * k ** 3 * ci * co
        total += us
        print("%2d -> %2d k%d: %6.1f us  %5.1f TFLOP/s" % (ci, co, k, us, flop / us / 1e6))
    print("sum of the nine: %.1f us" % total)
    for ia, ib in PAIRS:
        (ca, oa, ka), (cb, ob, kb) = SHAPES[ia], SHAPES[ib]

        def pair(ia=ia, ib=ib, ca=ca, cb=cb):
            _lib.check(lib.pcgc_train_conv_fwd_pair(plan, ia, ib, _lib.dptr(xs[ca]), _lib.dptr(xs[cb]), _lib.dptr(bs[ia]), _lib.dptr(bs[ib]),
                                                    _lib.dptr(ys[ia]), _lib.dptr(ys[ib]), B, D, 1, 1, _lib.stream()))
        us = timed(pair)
        flop = 2.0 * B * D ** 3 * (ka ** 3 * ca * oa + kb ** 3 * cb * ob)
        print("pair %2d -> %2d k%d | %2d -> %2d k%d: %6.1f us  %5.1f TFLOP/s" % (ca, oa, ka, cb, ob, kb, us, flop / us / 1e6))
    lib.pcgc_train_plan_destroy(plan)


if __name__ == "__main__":
    main()
