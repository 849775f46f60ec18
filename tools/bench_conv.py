"""Micro-benchmark of single convolution launches through pcgc_conv3d_fwd (GPU box only).
   python tools/bench_conv.py            # table of layer shapes x algorithms"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pcgcv1_amd.models import model_voxception as m

CASES = [  # cin, cout, k, stride, transposed, D, B
    (16, 4, 3, 1, False, 64, 8), (4, 8, 3, 1, False, 64, 8), (4, 4, 3, 1, False, 64, 8), (1, 16, 3, 1, False, 64, 8),
    (16, 1, 3, 1, False, 64, 8), (32, 8, 3, 1, False, 32, 8), (64, 16, 3, 1, False, 16, 8), (16, 32, 3, 1, False, 16, 8),
    (64, 16, 3, 1, False, 16, 64),
]


def main():
    rng = np.random.default_rng(0)
    for cin, cout, k, stride, tr, D, B in CASES:
        x = torch.from_numpy(rng.standard_normal((B, D, D, D, cin)).astype(np.float32)).cuda()
        ks = (k, k, k, cout, cin) if tr else (k, k, k, cin, cout)
        w = torch.from_numpy((rng.standard_normal(ks) * 0.1).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).cuda()
        ref = None
        row = []
        for algo in (1, 0, 3):
            try:
                y = m.conv3d(x, w, b, stride=stride, transposed=tr, relu=True, algo=algo)
            except Exception as e:
                row.append("algo%d: n/a" % algo)
                continue
            if ref is None:
                ref = y
            err = float((y - ref).abs().max())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                m.conv3d(x, w, b, stride=stride, transposed=tr, relu=True, algo=algo)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            dout = 2 * D if tr else D // stride
            flop = 2.0 * B * (D ** 3 if tr else dout ** 3) * k ** 3 * cin * cout
            row.append("algo%d: %.3f ms %.1f TF err %.1e" % (algo, ms, flop / ms / 1e9, err))
        print("cin=%d cout=%d k=%d s=%d t=%d D=%d B=%d | " % (cin, cout, k, stride, tr, D, B) + " | ".join(row))


if __name__ == "__main__":
    main()
