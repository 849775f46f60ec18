"""What would fusing the 64^3 kernels A and BC at workgroup granularity buy?  Measured pieces (PCGC_EXPERIMENTS=1 build):

  A  : all traffic | t12 stores dropped | t12 rows into an LDS ring (2 x ds_write_b128 per row) + one s_barrier per plane step
  BC : all traffic | t12 loads read nothing | t12 rows out of the LDS ring (ds_read_b128) + one s_barrier per plane step

The LDS variants are the SAME MFMA streams as the shipped kernels with the memory instructions a fused kernel would issue in
their place; a fused kernel additionally recomputes kernel A's halo rows ((R + 2) / R of its MFMAs for R rows per workgroup)
and has to balance the two phases over the workgroup's waves (DESIGN.md / profiles/HISTORY.md, round 5).
    PCGC_EXPERIMENTS=1 python -m pcgcv1_amd.build && gpurun -- python tools/exp/t_fuse_probe.py   (then rebuild without the variable)"""
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
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 1           # chunks visited in turn (1: the chunk stays in the Infinity Cache)
xs = [torch.rand(B * vox * 16, device=dev) for _ in range(NCH)]
ts = [torch.rand(B * vox * 8, device=dev) for _ in range(NCH)]


def run(which, abl, reps=60):
    def once(i):
        x, t = xs[i % NCH], ts[i % NCH]
        return f(x.data_ptr(), t.data_ptr(), x.data_ptr(), parr, B, which, 1, abl, st.cuda_stream)
    for i in range(10):
        assert once(i) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        once(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("us per launch of 8 cubes, %d chunk(s) in turn" % NCH)
for rep in range(3):
    a0, a1, a64 = run(0, 0), run(0, 1), run(0, 64)
    b0, b4, b64 = run(1, 0), run(1, 4), run(1, 64)
    print("A: all traffic %.1f | no t12 stores %.1f | t12 -> LDS ring + barrier %.1f      BC: all traffic %.1f | no t12 loads %.1f | t12 <- LDS ring + barrier %.1f"
          "      A + BC: %.1f now | %.1f traffic bound | %.1f with the LDS / barrier instructions (before halo recompute: x1.0625 on A at 32 rows, x1.125 at 16)"
          % (a0, a1, a64, b0, b4, b64, a0 + b0, a1 + b4, a64 + b64))
