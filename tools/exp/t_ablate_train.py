"""Training variants of the 64^3 row kernels (NDHWC): time of the forward pair with pieces of the traffic switched off.
    PCGC_EXPERIMENTS=1 python -m pcgcv1_amd.build && gpurun -- python tools/exp/t_ablate_train.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import _lib
lib = _lib.hip()
lib.pcgc_exp_set_vrn16_ablation.argtypes = [ctypes.c_int]
dev = _lib.require_gpu()
rng = np.random.default_rng(0)
shapes = [(27 * 16 * 4,), (4,), (27 * 4 * 8,), (8,), (16 * 4,), (4,), (27 * 4 * 4,), (4,), (4 * 8,), (8,)]
params = [torch.from_numpy((rng.standard_normal(s) * 0.05).astype(np.float32)).to(dev) for s in shapes]
parr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in params])
B, D, C = 8, 64, 16
x = torch.rand((B, D, D, D, C), device=dev)
t11, t21, t22 = (torch.empty((B, D, D, D, 4), device=dev) for _ in range(3))
pre, out = torch.empty_like(x), torch.empty_like(x)


def run(abl, reps=30):
    lib.pcgc_exp_set_vrn16_ablation(abl)
    def once():
        _lib.check(lib.pcgc_vrn_fwd_train(_lib.dptr(x), ctypes.cast(parr, ctypes.c_void_p), _lib.dptr(t11), _lib.dptr(t21), _lib.dptr(t22),
                                          _lib.dptr(pre), _lib.dptr(out), B, D, C, _lib.stream()))
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    lib.pcgc_exp_set_vrn16_ablation(0)
    return e0.elapsed_time(e1) / reps * 1e3


for rep in range(2):
    print("A + BC (training, NDHWC) us per 8 cubes: " + " | ".join("%s %.1f" % (w, run(a)) for a, w in
          ((0, "all traffic"), (32, "no pre stores"), (1, "no stores at all"), (2, "no residual loads"), (4, "no input loads"), (7, "no traffic"))))
