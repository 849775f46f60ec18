"""EntropyBottleneck (factorized prior) — operator surface of the reference's
models/entropy_model.py (class EntropyBottleneck, 8-306) on MI355X.

  __call__(inputs, training)  -> (values, likelihood)          entropy_model.py:153-181
  compress(inputs)            -> (string, min_v, max_v)        entropy_model.py:223-261
  decompress(strings, min_v, max_v, shape, channels=None)      entropy_model.py:263-306

The per-channel density network (72-98) and the likelihood (114-151) run in
`pcgc_factorized_likelihood` / `pcgc_factorized_pmf` (libpcgc_hip.so); the pmf
-> integer CDF step and the range coder are the sequential host tail
(libpcgc_host.so, coder_ops.py), one string for the whole batch as in the
reference.  Variable names keep the reference's spelling (`bais_i`, :58).
"""
import numpy as np
import torch

from .. import _lib, coder_ops


class EntropyBottleneck(object):
    def __init__(self, likelihood_bound=1e-9, range_coder_precision=16, init_scale=8, filters=(3, 3, 3)):
        self._likelihood_bound = float(likelihood_bound)
        self._range_coder_precision = int(range_coder_precision)
        self._init_scale = float(init_scale)
        self._filters = tuple(int(f) for f in filters)
        if self._filters != (3, 3, 3):
            raise NotImplementedError("the HIP density kernel is built for filters=(3,3,3) (the reference's default)")
        self.channels = None
        self.variables = None          # dict name -> numpy (checkpoint view)
        self._params = None            # flat device tensor in the order pcgc.h documents
        self._cdf_cache = {}
        self._pinned = {}              # stream -> (int16 symbols, int32 range) staging buffers of compress_async

    # -- variables -------------------------------------------------------
    def build(self, channels, rng=None):
        """entropy_model.py:25-70 initialisers."""
        rng = rng or np.random.default_rng(0)
        f = (1,) + self._filters + (1,)
        scale = self._init_scale ** (1.0 / (len(self._filters) + 1))
        v = {}
        for i in range(len(self._filters) + 1):
            init = np.log(np.expm1(1.0 / scale / f[i + 1]))
            v["matrix_%d" % i] = np.full((channels, f[i + 1], f[i]), init, np.float32)
            v["bais_%d" % i] = rng.uniform(-0.5, 0.5, (channels, f[i + 1], 1)).astype(np.float32)
            v["factor_%d" % i] = np.zeros((channels, f[i + 1], 1), np.float32)
        return self.load_weights(v, prefix="")

    def load_weights(self, weights, prefix="estimator"):
        p = prefix + "/" if prefix else ""
        names = ["%s_%d" % (k, i) for i in range(4) for k in ("matrix", "bais", "factor")]
        v = {}
        for n in names:
            arr = weights.get(p + n, weights.get(n))
            if arr is None:
                raise KeyError("missing entropy-bottleneck variable %s%s" % (p, n))
            v[n] = np.ascontiguousarray(arr, np.float32)
        self.channels = int(v["matrix_0"].shape[0])
        self.variables = v
        flat = np.concatenate([v[n].reshape(-1) for n in names])
        assert flat.size == self.channels * 44
        self._params = torch.from_numpy(flat).to(_lib.require_gpu())
        self._cdf_cache = {}
        return self

    def _ensure_built(self, channels):
        if self._params is None:
            self.build(int(channels))
        assert self.channels == int(channels), "channel mismatch"

    # -- forward ---------------------------------------------------------
    def __call__(self, inputs, training, noise=None):
        dev = _lib.require_gpu()
        x = inputs if torch.is_tensor(inputs) else torch.from_numpy(np.ascontiguousarray(inputs, np.float32))
        x = x.to(dev, torch.float32).contiguous()
        self._ensure_built(x.shape[-1])
        if training and noise is None:
            noise = torch.rand_like(x) - 0.5          # tf.random.uniform(-half, half), entropy_model.py:105-107
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        values = torch.empty_like(x)
        lik = torch.empty_like(x)
        _lib.check(_lib.hip().pcgc_factorized_likelihood(_lib.dptr(x), _lib.dptr(self._params),
                                                         _lib.dptr(noise) if training else None, _lib.dptr(values),
                                                         _lib.dptr(lik), x.numel(), self.channels,
                                                         self._likelihood_bound, _lib.stream()),
                   "pcgc_factorized_likelihood")
        return values, lik

    def _pmf(self, min_v, max_v):
        n = int(max_v) - int(min_v) + 1
        # called from whichever thread needs the table first (sharding's z-string thread among them): a fresh thread's
        # current device is 0, the variables live on the rank's device
        _lib.bind_device(self._params.device)
        pmf = torch.empty((self.channels, n), dtype=torch.float32, device=self._params.device)
        _lib.check(_lib.hip().pcgc_factorized_pmf(_lib.dptr(self._params), self.channels, int(min_v), int(max_v),
                                                  self._likelihood_bound, _lib.dptr(pmf), _lib.stream()),
                   "pcgc_factorized_pmf")
        return pmf.cpu().numpy()

    def _get_cdf(self, min_v, max_v):
        """entropy_model.py:183-221 -> int32 [1, C, N+1].  The table depends only on the (fixed) variables and
        the support, so it is cached per (min_v, max_v)."""
        key = (int(min_v), int(max_v))
        cdf = self._cdf_cache.get(key)
        if cdf is None:
            pmf = self._pmf(min_v, max_v)
            cdf = coder_ops.pmf_to_quantized_cdf(pmf, precision=self._range_coder_precision)
            cdf = cdf.reshape(1, self.channels, -1)
            self._cdf_cache[key] = cdf
        return cdf

    def quantize_minmax(self, x):
        """round-half-even + global min / max on the device -> (values tensor, min_v, max_v)."""
        q = torch.empty_like(x)
        mm = torch.empty(2, dtype=torch.int32, device=x.device)
        _lib.check(_lib.hip().pcgc_round_minmax(_lib.dptr(x), _lib.dptr(q), _lib.dptr(mm[0:1]), _lib.dptr(mm[1:2]),
                                                x.numel(), x.numel(), _lib.stream()), "pcgc_round_minmax")
        if x.numel() == 0:
            return q, 0, 0                 # nothing to code: an empty stream with the symbol range {0, 1}
        mn, mx = (int(v) for v in mm.cpu().numpy())
        return q, mn, mx

    def quantize_into(self, x, out):
        """round-half-even of x into the caller's buffer (what __call__(x, training=False) returns as values,
        entropy_model.py:161-163) — no likelihoods, no fresh tensor."""
        assert out.is_contiguous() and x.is_contiguous() and out.shape == x.shape
        if x.numel():
            mm = torch.empty(2, dtype=torch.int32, device=x.device)
            _lib.check(_lib.hip().pcgc_round_minmax(_lib.dptr(x), _lib.dptr(out), _lib.dptr(mm[0:1]), _lib.dptr(mm[1:2]),
                                                    x.numel(), x.numel(), _lib.stream()), "pcgc_round_minmax")
        return out

    def _values_from_symbols(self, sym, min_v, shape, dev):
        """decoded int16 symbols (host) -> float32 values on the device: upload + one kernel (sym + min_v)"""
        s16 = torch.from_numpy(sym).to(dev, non_blocking=False).reshape(-1)
        v = torch.empty(s16.numel(), dtype=torch.float32, device=dev)
        _lib.check(_lib.hip().pcgc_symbols_to_values(_lib.dptr(s16), int(min_v), _lib.dptr(v), s16.numel(), _lib.stream()),
                   "pcgc_symbols_to_values")
        return v.reshape(shape)

    def compress(self, inputs):
        dev = _lib.require_gpu()
        x = inputs if torch.is_tensor(inputs) else torch.from_numpy(np.ascontiguousarray(inputs, np.float32))
        x = x.to(dev, torch.float32).contiguous()
        self._ensure_built(x.shape[-1])
        values, min_v, max_v = self.quantize_minmax(x)
        if max_v == min_v:
            max_v += 1      # 1-symbol pmf cannot be quantised (entropy_model.py:192-193 TODO): code with a 2-symbol support
        cdf = self._get_cdf(min_v, max_v)
        sym = (values.reshape(-1, self.channels).to(torch.int32) - min_v).to(torch.int16).cpu().numpy()
        strings = coder_ops.range_encode(sym, cdf, precision=self._range_coder_precision)
        return strings, min_v, max_v

    def compress_async(self, inputs):
        """compress() with the sequential range coding on a helper thread.  Returns a callable that joins the
        thread and gives (string, min_v, max_v); the device part (rounding, range, symbols to host: ONE round trip on the
        current stream) runs now."""
        import threading
        dev = _lib.require_gpu()
        x = inputs if torch.is_tensor(inputs) else torch.from_numpy(np.ascontiguousarray(inputs, np.float32))
        x = x.to(dev, torch.float32).contiguous()
        self._ensure_built(x.shape[-1])
        if x.numel() == 0:
            values, min_v, max_v = self.quantize_minmax(x)
            vals = np.zeros((0, self.channels), np.int16)
        else:
            q = torch.empty(x.numel(), dtype=torch.int16, device=x.device)     # the coder's input type, straight from the kernel
            mm = torch.empty(2, dtype=torch.int32, device=x.device)
            _lib.check(_lib.hip().pcgc_round_minmax_i16(_lib.dptr(x), _lib.dptr(q), _lib.dptr(mm[0:1]), _lib.dptr(mm[1:2]),
                                                        x.numel(), x.numel(), _lib.stream()), "pcgc_round_minmax_i16")
            skey = int(torch.cuda.current_stream().cuda_stream)
            hv = self._pinned.get(skey)
            if hv is None or hv[0].numel() < x.numel():
                hv = self._pinned[skey] = (torch.empty(x.numel(), dtype=torch.int16, pin_memory=True),
                                           torch.empty(2, dtype=torch.int32, pin_memory=True), [None])
            if hv[2][0] is not None:
                hv[2][0].wait()            # the previous call's coder thread has taken its copy of the staging buffer
            hv[0][:x.numel()].copy_(q, non_blocking=True)
            hv[1].copy_(mm, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            min_v, max_v = int(hv[1][0]), int(hv[1][1])
            if min_v < -32768 or max_v > 32767 or max_v - min_v > 32767:
                raise OverflowError("hyperprior symbols %d..%d do not fit 16 bits" % (min_v, max_v))
            vals = hv[0][:x.numel()].numpy().reshape(-1, self.channels)
        if max_v == min_v:
            max_v += 1
        cdf = self._get_cdf(min_v, max_v)
        box = {}
        copied = threading.Event()
        if x.numel():
            hv[2][0] = copied

        def work():
            try:
                try:                                    # straight from the staging buffer (values - min_v inside the coder)
                    box["s"] = coder_ops.range_encode_values(vals, min_v, cdf, precision=self._range_coder_precision)
                finally:
                    copied.set()                        # ... which the next call may now reuse
            except Exception as e:          # surfaced by the join below
                box["e"] = e
        th = _lib.workers().submit(work)

        def join():
            th.result()
            if "e" in box:
                raise box["e"]
            return box["s"], min_v, max_v
        return join

    def decompress_async(self, strings, min_v, max_v, shape, channels=None):
        """decompress() with the sequential decoding on a helper thread.  Returns part(lo, hi) -> float32 device tensor
        of the first-dimension slice [lo:hi] (cubes lo..hi-1), available as soon as those symbols are decoded."""
        dev = _lib.require_gpu()
        shape = tuple(int(s) for s in shape)
        self._ensure_built(channels if channels is not None else shape[-1])
        cdf = self._get_cdf(int(min_v), int(max_v))
        rows = int(np.prod(shape)) // self.channels
        per = rows // max(shape[0], 1)                       # rows per cube
        sym, wait = coder_ops.range_decode_async(strings, (rows, self.channels), cdf, precision=self._range_coder_precision)

        def part(lo, hi):
            _lib.mark("z wait [%d:%d]" % (lo, hi))
            wait(hi * per)
            _lib.mark("z ready [%d:%d]" % (lo, hi))
            return self._values_from_symbols(sym[lo * per:hi * per], min_v, (hi - lo,) + shape[1:], dev)
        part.ready = lambda hi: wait(hi * per, block=False)        # True when cubes [0, hi) are decoded
        return part

    def decompress(self, strings, min_v, max_v, shape, channels=None):
        dev = _lib.require_gpu()
        shape = tuple(int(s) for s in shape)
        self._ensure_built(channels if channels is not None else shape[-1])
        cdf = self._get_cdf(int(min_v), int(max_v))
        rows = int(np.prod(shape)) // self.channels
        sym = coder_ops.range_decode(strings, (rows, self.channels), cdf, precision=self._range_coder_precision)
        # int16 symbols up, offset and float conversion on the device (three host passes over millions of symbols otherwise)
        return self._values_from_symbols(sym, min_v, shape, dev)
