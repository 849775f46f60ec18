"""Is kernel A on compact wave tiles (tools/exp/exp_seg_a.hip, variant 3: the row kernel's three kw partial sums formed on shifted INPUTS) the
same function, bit for bit, as the product's vrn16a_row_kernel?  Then the empty-cube responses, the goldens and the dense launches stay
valid when the analysis' skipping launches move to compact tiles.  GPU box:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSEG_A_SHARED tools/exp/exp_seg_a.hip -o tools/exp/_build/libexp_seg_a.so
    python tools/exp/t_seg_a_bits.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pcgcv1_amd import _lib

root = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(root, "_build", "libexp_seg_a.so"))
probe.seg_a_launch.restype = ctypes.c_int
probe.seg_a_launch.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib, dev = _lib.hip(), _lib.require_gpu()
g = torch.Generator(device="cpu").manual_seed(5)
B, D, C = 3, 64, 16
shapes = [(3, 3, 3, C, 4), (4,), (3, 3, 3, 4, 8), (8,), (1, 1, 1, C, 4), (4,), (3, 3, 3, 4, 4), (4,), (1, 1, 1, 4, 8), (8,)]
params = [(torch.randn(sh, generator=g) * (0.15 if len(sh) > 1 else 0.05)).to(dev) for sh in shapes]
arr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in params])
x = torch.relu(torch.randn((B, D, D, D, C), generator=g)).to(dev)
x[:, :, :, 20:40] = 0                                         # an empty slab: exact zeros in the input, as in the analysis
vox = B * D * D * D
ws = torch.zeros(int(lib.pcgc_vrn_workspace_bytes(B, D, C)), dtype=torch.uint8, device=dev)
out = torch.empty_like(x)
_lib.check(lib.pcgc_vrn_fwd(_lib.dptr(x), ctypes.cast(arr, ctypes.c_void_p), _lib.dptr(out), B, D, C, _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_vrn_fwd")
base = (ws.data_ptr() + 255) & ~255
off = (base - ws.data_ptr()) // 4
t12_ref = ws.view(torch.float32)[off + vox * C: off + vox * C + vox * 8].clone()         # what vrn16a_row_kernel wrote (Q4, 2 quads)
xq = torch.empty(vox * C, dtype=torch.float32, device=dev)
_lib.check(lib.pcgc_layout_q4(_lib.dptr(x), _lib.dptr(xq), B, D, C, 1, _lib.stream()), "pcgc_layout_q4")
for var in (1, 2, 3):
    t12 = torch.full((vox * 8,), float("nan"), dtype=torch.float32, device=dev)
    rc = probe.seg_a_launch(xq.data_ptr(), t12.data_ptr(), params[0].data_ptr(), params[1].data_ptr(), params[4].data_ptr(), params[5].data_ptr(),
                            B, var, _lib.stream())
    torch.cuda.synchronize()
    assert rc == 0
    same = torch.equal(t12, t12_ref)
    print("variant %d: %s the row kernel's output (%d of %d values differ, max |diff| %.3g)" % (
        var, "bit-identical to" if same else "NOT the bits of", int((t12 != t12_ref).sum()), t12.numel(), float((t12 - t12_ref).abs().max())))

# ---- kernel BC on the same tiles (tools/exp/exp_seg_bc.hip): out against pcgc_vrn_fwd's, and its time
pbc = ctypes.CDLL(os.path.join(root, "_build", "libexp_seg_bc.so"))
pbc.seg_bc_launch.restype = ctypes.c_int
pbc.seg_bc_launch.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out_ref_q4 = torch.empty(vox * C, dtype=torch.float32, device=dev)
_lib.check(lib.pcgc_layout_q4(_lib.dptr(out), _lib.dptr(out_ref_q4), B, D, C, 1, _lib.stream()), "pcgc_layout_q4")


def bc(t12_in, xq_in, nb, ld, out_t):
    return pbc.seg_bc_launch(t12_in.data_ptr(), xq_in.data_ptr(), out_t.data_ptr(), params[2].data_ptr(), params[3].data_ptr(), params[6].data_ptr(),
                             params[7].data_ptr(), params[8].data_ptr(), params[9].data_ptr(), nb, ld, _lib.stream())
for ld in (8, 4, 16):
    o = torch.full((vox * C,), float("nan"), dtype=torch.float32, device=dev)
    assert bc(t12_ref, xq, B, ld, o) == 0
    torch.cuda.synchronize()
    print("BC, %2d planes per wave: %s the row kernels' block output (%d of %d values differ, max |diff| %.3g)" % (
        ld, "bit-identical to" if torch.equal(o, out_ref_q4) else "NOT the bits of", int((o != out_ref_q4).sum()), o.numel(), float((o - out_ref_q4).abs().max())))
for nb in (8, 16):
    xs = torch.relu(torch.randn((nb * D * D * D * C,), generator=g)).to(dev)
    ts = torch.relu(torch.randn((nb * D * D * D * 8,), generator=g)).to(dev)
    o = torch.empty_like(xs)
    line = []
    for ld in (8, 4, 16):
        for _ in range(3):
            bc(ts, xs, nb, ld, o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            bc(ts, xs, nb, ld, o)
        e1.record()
        torch.cuda.synchronize()
        line.append("LD %d %.1f us" % (ld, 1e3 * e0.elapsed_time(e1) / 20))
    print("BC on compact tiles, B = %2d cubes: %s   (vrn16bc_row_kernel: 66.5 us per 8 cubes dense)" % (nb, ", ".join(line)))
