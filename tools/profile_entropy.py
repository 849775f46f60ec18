"""Fine-grained wall-clock of the entropy path (GPU box): python tools/profile_entropy.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcgcv1_amd import _lib, checkpoint, process, synthetic, transform, coder_ops
from pcgcv1_amd.models import model_voxception as model

w = synthetic.make_weights(seed=1300, profile=sys.argv[1] if len(sys.argv) > 1 else "sparse")
checkpoint._CACHE["bench"] = w
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
c = transform.get_codec(model, "bench")
for _ in range(2):
    out = transform.compress_hyper(cubes, model, "bench")
    transform.decompress_hyper(*out, model, "bench")
ys = c.analysis_transform(cubes); zs = c.hyper_encoder(ys)
zh, _ = c.entropy_bottleneck(zs, False)
locs, scales = c.hyper_decoder(zh, lower_bound=1e-9)
sc, eb = c.conditional_entropy_model, c.entropy_bottleneck
def T(f, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
B = ys.shape[0]; rows = ys.numel(); seg = rows // B
print("z compress total %.2f ms" % T(lambda: eb.compress(zs))[0])
t, (values, mn, mx) = T(lambda: eb.quantize_minmax(zs)); print("  z quantize_minmax %.2f" % t)
t, cdf = T(lambda: eb._get_cdf(mn, mx)); print("  z get_cdf %.2f" % t)
t, sym = T(lambda: (values.reshape(-1, 8).to(torch.int32) - mn).to(torch.int16).cpu().numpy()); print("  z sym to host %.2f" % t)
t, s = T(lambda: coder_ops.range_encode(sym, cdf)); print("  z range_encode %.2f (%d syms, %d bytes)" % (t, sym.size, len(s)))
t, d = T(lambda: coder_ops.range_decode(s, sym.shape, cdf)); print("  z range_decode %.2f" % t)
print("y compress_cubes total %.2f ms" % T(lambda: sc.compress_cubes(ys, locs, scales))[0])
t, (yh, mnd, mxd) = T(lambda: sc.quantize_minmax(ys, B)); print("  y quantize_minmax %.2f" % t)
mn_, mx_ = mnd.cpu().numpy(), mxd.cpu().numpy(); ncols = int((mx_ - mn_).max()) + 1
lohi = torch.empty(rows, dtype=torch.int32, device=ys.device)
t, _ = T(lambda: _lib.check(_lib.hip().pcgc_laplace_cdf(_lib.dptr(locs), _lib.dptr(scales), _lib.dptr(mnd), _lib.dptr(mxd), rows, seg, ncols, 1e-9, _lib.dptr(yh), None, _lib.dptr(lohi), _lib.stream()))); print("  y cdf kernel (lohi) %.2f  ncols=%d" % (t, ncols))
hl = sc._pin("lohi", (rows,), torch.int32)
t, _ = T(lambda: hl.copy_(lohi, non_blocking=True)); print("  y lohi D2H %.2f (%.0f MB)" % (t, rows * 4 / 1e6))
cap = seg * 2 + 1024; outb = np.empty((B, cap), np.uint8); lens = np.zeros(B, np.int64)
for nt in (256, 64, 16):
    t, _ = T(lambda: _lib.check_host(_lib.host().pcgc_range_encode_lohi_batch(hl.data_ptr(), B, seg, 16, _lib.nptr(outb), cap, _lib.nptr(lens), nt))); print("  y host encode %d threads %.2f" % (nt, t))
t, _ = T(lambda: [outb[i, :lens[i]].tobytes() for i in range(B)]); print("  y slice to bytes %.2f" % t)
t, _ = T(lambda: np.empty((B, cap), np.uint8)); print("  alloc out %.2f" % t)
ystr, ymn, ymx = sc.compress_cubes(ys, locs, scales)
print("y decompress_cubes total %.2f ms" % T(lambda: sc.decompress_cubes(ystr, locs, scales, ymn, ymx, (1, 16, 16, 16, 16)))[0])
cdfd = torch.empty((rows, ncols), dtype=torch.int16, device=ys.device)
t, _ = T(lambda: _lib.check(_lib.hip().pcgc_laplace_cdf(_lib.dptr(locs), _lib.dptr(scales), _lib.dptr(mnd), _lib.dptr(mxd), rows, seg, ncols, 1e-9, None, _lib.dptr(cdfd), None, _lib.stream()))); print("  y cdf kernel (rows) %.2f" % t)
hc = sc._pin("cdf", (rows, ncols), torch.int16)
t, _ = T(lambda: hc.copy_(cdfd, non_blocking=True)); print("  y cdf D2H %.2f (%.0f MB)" % (t, rows * ncols * 2 / 1e6))
