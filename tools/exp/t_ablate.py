"""Where do the 64^3 row kernels spend their time?  The same instruction stream with memory traffic removed piece by
piece (zero-record descriptors: the loads / stores still issue, nothing moves).
    PCGC_EXPERIMENTS=1 python -m pcgcv1_amd.build && gpurun -- python tools/exp/t_ablate.py   (then rebuild without the variable)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import _lib
lib = _lib.hip()
f = lib.pcgc_exp_vrn16_row
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = _lib.require_gpu()
rng = np.random.default_rng(0)
shapes = [(27 * 16 * 4,), (4,), (27 * 4 * 8,), (8,), (16 * 4,), (4,), (27 * 4 * 4,), (4,), (4 * 8,), (8,)]
params = [torch.from_numpy((rng.standard_normal(s) * 0.05).astype(np.float32)).to(dev) for s in shapes]
parr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in params])
vox = 64 ** 3
st = torch.cuda.current_stream()
B = 8
x, t = torch.rand(B * vox * 16, device=dev), torch.rand(B * vox * 8, device=dev)


def run(which, abl, reps=60):
    for _ in range(10):
        assert f(x.data_ptr(), t.data_ptr(), x.data_ptr(), parr, B, which, 1, abl, st.cuda_stream) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f(x.data_ptr(), t.data_ptr(), x.data_ptr(), parr, B, which, 1, abl, st.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rep in range(2):
    for which, name in ((0, "A"), (1, "BC")):
        row = []
        for abl, what in ((0, "all traffic"), (1, "no stores"), (2, "no residual"), (4, "no input loads"), (6, "no loads"), (7, "no traffic")):
            if which == 0 and abl in (2, 6):
                continue
            row.append("%s %.1f" % (what, run(which, abl)))
        if which == 1:
            for abl, what in ((8, "residual loads nt"), (16, "stores nt"), (24, "both nt")):
                row.append("%s %.1f" % (what, run(which, abl)))
        print(name, "us per 8 cubes:", " | ".join(row))
