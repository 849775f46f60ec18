"""Per-symbol time of pcgc_range_decode_u16_batch on one thread (host only): 6-symbol Laplace-like rows, 65 536 symbols per stream."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from pcgcv1_amd import _lib, coder_ops
rng = np.random.default_rng(0)
rows, N = 65536, 6
# skewed pmf per row (Laplace-like), quantised to 16-bit CDFs through the product's own quantiser
loc = rng.uniform(0.5, 2.5, rows); sc = rng.uniform(0.15, 0.9, rows)
k = np.arange(N)[None, :]
pmf = np.exp(-np.abs(k - loc[:, None]) / sc[:, None]).astype(np.float32)
pmf /= pmf.sum(1, keepdims=True)
cdf = coder_ops.pmf_to_quantized_cdf(pmf)                       # int32 [rows, N+1]
u = rng.random(rows) * 65536
sym = np.minimum((cdf[:, 1:] <= u[:, None]).sum(1), N - 1).astype(np.int16)
enc = coder_ops.range_encode(sym.reshape(rows, 1), cdf.reshape(rows, 1, N + 1))
print("bytes", len(enc), "bits/sym", 8 * len(enc) / rows)
host = _lib.host()
S = 8                                                          # streams (threads = 1: per-stream latency is what matters)
blob = np.frombuffer(enc * S + b"\0", np.uint8)
offsets = (np.arange(S) * len(enc)).astype(np.int64); lens = np.full(S, len(enc), np.int64)
rows16 = np.tile(cdf[:, :N].astype(np.uint16), (S, 1))        # lower bounds, ncols = N
n_sym = np.full(S, N, np.int32)
out = np.empty(S * rows, np.int16)
for rep in range(3):
    t = time.perf_counter()
    rc = host.pcgc_range_decode_u16_batch(_lib.nptr(blob), _lib.nptr(offsets), _lib.nptr(lens), S, rows, _lib.nptr(rows16), N, _lib.nptr(n_sym), 16, _lib.nptr(out), 1)
    dt = time.perf_counter() - t
    assert rc == 0 and np.array_equal(out[:rows], sym) and np.array_equal(out[-rows:], sym)
    print("decode %.2f ns / symbol" % (1e9 * dt / (S * rows)))
