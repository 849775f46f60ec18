"""The segment ("gather") kernels of csrc/vrn_seg.hip against the row kernels of csrc/vrn_row.hip, stand-alone (GPU box):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPCGC_SEG_PROBE -I include pcgcv1_amd/csrc/vrn_seg.hip -o tools/exp/_build/libseg_probe.so
    python tools/exp/t_seg_gather.py
1. every slot, natural order: kernel A's tensor1_1 | tensor2_1 and kernel BC's block output bit-identical to what pcgc_vrn_fwd's row
   kernels wrote;  2. a random subset of the slots in random order: exactly those slots written, with the same bits;  3. slots of the
   inputs marked "not written": the kernels read the empty-cube responses there (per-lane select) — against the kernels run on
   materialised inputs;  4. time per launch, dense, 8 and 16 cubes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pcgcv1_amd import _lib

root = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(root, "_build", "libseg_probe.so"))
probe.seg_probe_launch.restype = ctypes.c_int
u32, vp = ctypes.c_uint, ctypes.c_void_p
probe.seg_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, vp, u32, u32, u32, u32, u32, vp, vp, ctypes.c_int, vp, vp, vp, vp]
lib, dev = _lib.hip(), _lib.require_gpu()
g = torch.Generator(device="cpu").manual_seed(5)
D, C = 64, 16
V = D * D * D
shapes = [(3, 3, 3, C, 4), (4,), (3, 3, 3, 4, 8), (8,), (1, 1, 1, C, 4), (4,), (3, 3, 3, 4, 4), (4,), (1, 1, 1, 4, 8), (8,)]
params = [(torch.randn(sh, generator=g) * (0.15 if len(sh) > 1 else 0.05)).to(dev) for sh in shapes]
arr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in params])
PAD = 2 << 20


def q4(t, B, ch):           # NDHWC [B, D, D, D, ch] -> flat Q4
    o = torch.empty(B * V * ch, dtype=torch.float32, device=dev)
    _lib.check(lib.pcgc_layout_q4(_lib.dptr(t.contiguous()), _lib.dptr(o), B, D, ch, 1, _lib.stream()), "pcgc_layout_q4")
    return o


def row_reference(x):       # x NDHWC -> (t12 Q4 as the row kernel A wrote it, out Q4)
    B = x.shape[0]
    ws = torch.zeros(int(lib.pcgc_vrn_workspace_bytes(B, D, C)), dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    _lib.check(lib.pcgc_vrn_fwd(_lib.dptr(x), ctypes.cast(arr, vp), _lib.dptr(out), B, D, C, _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_vrn_fwd")
    base = (ws.data_ptr() + 255) & ~255
    off = (base - ws.data_ptr()) // 4
    return ws.view(torch.float32)[off + B * V * C: off + B * V * C + B * V * 8].clone(), q4(out, B, C)


class Window(object):
    """[pad][x: B cubes x 16 ch][t12: B x 8][out: B x 16][Ex: 1 x 16][Et: 1 x 8] as one allocation; offsets in bytes from its start"""
    def __init__(self, B):
        self.B = B
        n = PAD // 4 + B * V * (16 + 8 + 16) + V * (16 + 8)
        self.buf = torch.full((n,), float("nan"), dtype=torch.float32, device=dev)
        self.x_off = PAD
        self.t_off = self.x_off + B * V * 64
        self.out_off = self.t_off + B * V * 32
        self.ex_off = self.out_off + B * V * 64
        self.et_off = self.ex_off + V * 64

    def view(self, off, n):
        return self.buf[off // 4: off // 4 + n]

    def launch(self, which, slots, in_virt=None, res_virt=None, ein=0, eres=0, nonneg=1, out_off=None, max_slots=None):
        sl = torch.as_tensor(np.asarray(slots, np.uint32).view(np.int32)).to(dev)
        n = torch.tensor([len(slots)], dtype=torch.int32, device=dev)
        rc = probe.seg_probe_launch(which, nonneg, self.buf.data_ptr(), self.x_off, self.t_off, self.out_off if out_off is None else out_off, ein, eres,
                                    sl.data_ptr(), n.data_ptr(), max_slots or max(len(slots), 1), in_virt.data_ptr() if in_virt is not None else None,
                                    res_virt.data_ptr() if res_virt is not None else None, ctypes.cast(arr, vp), _lib.stream())
        torch.cuda.synchronize()
        assert rc == 0
        self._keep = (sl, n)


def slot_mask(B, slots, ch):
    """bool Q4 mask [B * V * ch] of the voxels of the given slots"""
    m = np.zeros((B, 8, 8, 32, 2, ch // 4, 4, 16, 4), bool)          # b, dt, plane, ht, row, quad, seg, voxel, lane-channel
    s = np.asarray(slots, np.int64)
    m[s >> 10, (s >> 7) & 7, :, (s >> 2) & 31, :, :, s & 3] = True
    return torch.from_numpy(m.reshape(-1)).to(dev)


def same(a, b):
    return bool(torch.equal(a, b)) or bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


B = 3
x = torch.relu(torch.randn((B, D, D, D, C), generator=g)).to(dev)
x[:, :, :, 20:40] = 0
t12_ref, out_ref = row_reference(x)
w = Window(B)
w.view(w.x_off, B * V * 16).copy_(q4(x, B, 16))
allslots = np.arange(B * 1024, dtype=np.uint32)
ok = True
# ---- 1. dense, natural order
w.launch(0, allslots)
r = same(w.view(w.t_off, B * V * 8), t12_ref); ok &= r
print("kernel A, every slot in natural order: %s the row kernel's tensor1_1 | tensor2_1 (%d of %d values differ)" % (
    "bit-identical to" if r else "NOT", int((w.view(w.t_off, B * V * 8) != t12_ref).sum()), B * V * 8))
w.launch(1, allslots)
r = same(w.view(w.out_off, B * V * 16), out_ref); ok &= r
print("kernel BC, every slot in natural order: %s the row kernels' block output (%d of %d values differ)" % (
    "bit-identical to" if r else "NOT", int((w.view(w.out_off, B * V * 16) != out_ref).sum()), B * V * 16))
# in place (out = x), as the transforms run it
xq = w.view(w.x_off, B * V * 16).clone()
w.launch(1, allslots, out_off=w.x_off)
r = same(w.view(w.x_off, B * V * 16), out_ref); ok &= r
print("kernel BC in place on the block input: %s" % ("bit-identical" if r else "DIFFERS"))
w.view(w.x_off, B * V * 16).copy_(xq)
# ---- 2. a random subset in random order (the last wave is not full)
rng = np.random.default_rng(3)
sub = rng.permutation(allslots)[: int(0.31 * len(allslots)) + 1]
for which, off, ch, ref in ((0, w.t_off, 8, t12_ref), (1, w.out_off, 16, out_ref)):
    if which == 1:
        w.view(w.t_off, B * V * 8).copy_(t12_ref)
    w.view(off, B * V * ch).fill_(float("nan"))
    w.launch(which, sub, max_slots=len(allslots))
    got, m = w.view(off, B * V * ch), slot_mask(B, sub, ch)
    r = bool(torch.equal(got[m], ref[m])) and bool(torch.isnan(got[~m]).all()); ok &= r
    print("kernel %s, %d random slots in random order: %s" % ("A" if which == 0 else "BC", len(sub), "those slots written, same bits" if r else "WRONG"))
# ---- 3. inputs with slots that were "not written": the empty-cube responses stand in, per lane
ex = torch.relu(torch.randn((1, D, D, D, 16), generator=g)).to(dev)
et = torch.relu(torch.randn((1, D, D, D, 8), generator=g)).to(dev)
w.view(w.ex_off, V * 16).copy_(q4(ex, 1, 16))
w.view(w.et_off, V * 8).copy_(q4(et, 1, 8))
virt_x = torch.from_numpy(rng.integers(0, 16, B * 256, dtype=np.uint8)).to(dev)
virt_t = torch.from_numpy(rng.integers(0, 16, B * 256, dtype=np.uint8)).to(dev)


def unwritten(v):        # table -> slot codes marked not written
    v = v.cpu().numpy()
    return np.array([i * 4 + s for i in range(len(v)) for s in range(4) if (v[i] >> s) & 1], np.uint32)


mx, mt = slot_mask(B, unwritten(virt_x), 16), slot_mask(B, unwritten(virt_t), 8)
x_true = w.view(w.x_off, B * V * 16).clone()
x_eff = torch.where(mx, w.view(w.ex_off, V * 16).repeat(B), x_true)
t_eff = torch.where(mt, w.view(w.et_off, V * 8).repeat(B), t12_ref)
# references: the kernels themselves on materialised inputs
w.view(w.x_off, B * V * 16).copy_(x_eff)
w.launch(0, allslots)
a_ref = w.view(w.t_off, B * V * 8).clone()
w.view(w.t_off, B * V * 8).copy_(t_eff)
w.launch(1, allslots)
bc_ref = w.view(w.out_off, B * V * 16).clone()
# now with garbage in the unwritten slots and the tables
w.view(w.x_off, B * V * 16).copy_(torch.where(mx, torch.full_like(x_true, float("nan")), x_true))
w.view(w.t_off, B * V * 8).fill_(float("nan"))
w.launch(0, allslots, in_virt=virt_x, ein=w.ex_off)
r = same(w.view(w.t_off, B * V * 8), a_ref); ok &= r
print("kernel A with %d input slots not written (read from the empty-cube response): %s" % (int(len(unwritten(virt_x))), "same bits" if r else "WRONG (%d differ)" % int((w.view(w.t_off, B * V * 8) != a_ref).sum())))
w.view(w.t_off, B * V * 8).copy_(torch.where(mt, torch.full_like(t_eff, float("nan")), t12_ref))
w.view(w.out_off, B * V * 16).fill_(float("nan"))
w.launch(1, allslots, in_virt=virt_t, res_virt=virt_x, ein=w.et_off, eres=w.ex_off)
r = same(w.view(w.out_off, B * V * 16), bc_ref); ok &= r
print("kernel BC with %d tensor1_1 | tensor2_1 slots and %d residual slots not written: %s" % (
    len(unwritten(virt_t)), len(unwritten(virt_x)), "same bits" if r else "WRONG (%d differ)" % int((w.view(w.out_off, B * V * 16) != bc_ref).sum())))
# ---- 4. time, dense
for nb in (8, 16):
    ww = Window(nb)
    ww.buf.normal_(generator=None).relu_()
    sl = np.arange(nb * 1024, dtype=np.uint32)
    sld = torch.as_tensor(sl.view(np.int32)).to(dev)
    for frac in (1.0, 0.5, 0.25):
        n = torch.tensor([int(frac * len(sl))], dtype=torch.int32, device=dev)
        line = []
        for which in (0, 1):
            def go():
                return probe.seg_probe_launch(which, 1, ww.buf.data_ptr(), ww.x_off, ww.t_off, ww.x_off, 0, 0, sld.data_ptr(), n.data_ptr(), len(sl), None, None,
                                              ctypes.cast(arr, vp), _lib.stream())
            for _ in range(3):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                go()
            e1.record()
            torch.cuda.synchronize()
            line.append("%s %.1f us" % ("A" if which == 0 else "BC", 1e3 * e0.elapsed_time(e1) / 20))
        print("%2d cubes, the first %3.0f %% of the slots: %s   (row kernels, dense: A 72-73 us, BC 65-67 us per 8 cubes)" % (nb, 100 * frac, ", ".join(line)))
print("ALL SAME" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
