"""SymmetricConditional (Laplace prior conditioned on loc / scale) — operator surface of
the reference's models/conditional_entropy_model.py (class SymmetricConditional, 8-201).

  __call__(inputs, loc, scale, training) -> (values, likelihood)      :71-93
  compress(inputs, loc, scale)           -> (string, min_v, max_v)     :126-163
  decompress(strings, loc, scale, min_v, max_v, datashape)            :165-201

plus the batched forms compress_hyper / decompress_hyper use instead of the reference's
per-cube tf.map_fn (transform.py:157-168, 238-248):

  compress_cubes(ys, locs, scales)  -> (list of strings, min_vs, max_vs)   one string per cube
  decompress_cubes(strings, locs, scales, min_vs, max_vs, datashape)

Device side (libpcgc_hip.so): rounding + per-cube min/max, the Laplace pmf table and its
16-bit quantised CDF for every (voxel, channel) row.  Host side (libpcgc_host.so): the
sequential range coder, one thread per cube stream.
"""
import numpy as np
import torch

from .. import _lib

_MAX_SYMBOLS = 32


class SymmetricConditional(object):
    def __init__(self, likelihood_bound=1e-9, range_coder_precision=16):
        self._likelihood_bound = float(likelihood_bound)
        self._range_coder_precision = int(range_coder_precision)
        if self._range_coder_precision != 16:
            raise NotImplementedError("the device CDF kernel emits 16-bit CDFs (the reference's only setting)")
        self._pinned = {}

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _dev(t):
        dev = _lib.require_gpu()
        if not torch.is_tensor(t):
            t = torch.from_numpy(np.ascontiguousarray(t, np.float32))
        return t.to(dev, torch.float32).contiguous()

    def _pin(self, key, shape, dtype):
        n = int(np.prod(shape))
        buf = self._pinned.get(key)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
            self._pinned[key] = buf
        return buf[:n].view(*shape)

    # -- forward ---------------------------------------------------------
    def __call__(self, inputs, loc, scale, training, noise=None):
        x, loc, scale = self._dev(inputs), self._dev(loc), self._dev(scale)
        if training and noise is None:
            noise = torch.rand_like(x) - 0.5          # conditional_entropy_model.py:62-64
        if noise is not None:
            noise = self._dev(noise)
        values, lik = torch.empty_like(x), torch.empty_like(x)
        _lib.check(_lib.hip().pcgc_laplace_likelihood(_lib.dptr(x), _lib.dptr(loc), _lib.dptr(scale),
                                                      _lib.dptr(noise) if training else None, _lib.dptr(values),
                                                      _lib.dptr(lik), x.numel(), self._likelihood_bound, _lib.stream()),
                   "pcgc_laplace_likelihood")
        return values, lik

    # -- encode ----------------------------------------------------------
    def quantize_minmax(self, ys, n_seg):
        """Round + per-segment min/max on the device. Returns (y_hat, seg_min, seg_max int32 device tensors)."""
        q = torch.empty_like(ys)
        mn = torch.empty(n_seg, dtype=torch.int32, device=ys.device)
        mx = torch.empty(n_seg, dtype=torch.int32, device=ys.device)
        _lib.check(_lib.hip().pcgc_round_minmax(_lib.dptr(ys), _lib.dptr(q), _lib.dptr(mn), _lib.dptr(mx), ys.numel(),
                                                ys.numel() // n_seg, _lib.stream()), "pcgc_round_minmax")
        return q, mn, mx

    def _check_range(self, mn, mx):
        n = int((mx - mn).max()) + 1
        if n > _MAX_SYMBOLS:
            raise ValueError("symbol range of %d values exceeds %d (the container stores |min|,|max| <= 15, "
                             "inout_bitstream.py:95-96)" % (n, _MAX_SYMBOLS))
        assert int((mx - mn).min()) >= 1
        return n

    @staticmethod
    def _widen(mn, mx):
        """A cube whose symbols are all equal gives a 1-symbol pmf, which pmf_to_quantized_cdf cannot
        quantise (the reference notes this as an unhandled TODO, entropy_model.py:192-193, and would fail).
        Such cubes get a 2-symbol support instead: max+1 (or min-1 at the container's upper limit 15); the
        range travels in the header, so the decoder builds the same CDF."""
        same = mx == mn
        if same.any():
            mx = np.where(same & (mx < 15), mx + 1, mx)
            mn = np.where(same & (mx == mn), mn - 1, mn)
        return mn.astype(np.int32), mx.astype(np.int32)

    def compress_cubes(self, ys, locs, scales, n_threads=None):
        ys, locs, scales = self._dev(ys), self._dev(locs), self._dev(scales)
        B = int(ys.shape[0])
        if B == 0:
            return [], np.zeros(0, np.int32), np.zeros(0, np.int32)
        rows = ys.numel()
        seg = rows // B
        y_hat, mn_d, mx_d = self.quantize_minmax(ys, B)
        mn0, mx0 = mn_d.cpu().numpy(), mx_d.cpu().numpy()
        mn, mx = self._widen(mn0, mx0)
        if not (np.array_equal(mn, mn0) and np.array_equal(mx, mx0)):
            mn_d, mx_d = torch.from_numpy(mn).to(ys.device), torch.from_numpy(mx).to(ys.device)
        ncols = self._check_range(mn, mx)
        lohi = torch.empty(rows, dtype=torch.int32, device=ys.device)
        _lib.check(_lib.hip().pcgc_laplace_cdf(_lib.dptr(locs), _lib.dptr(scales), _lib.dptr(mn_d), _lib.dptr(mx_d), rows,
                                               seg, ncols, self._likelihood_bound, _lib.dptr(y_hat), None,
                                               _lib.dptr(lohi), _lib.stream()), "pcgc_laplace_cdf")
        host_lohi = self._pin("lohi", (rows,), torch.int32)
        host_lohi.copy_(lohi, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        cap = seg * 2 + 1024
        out = np.empty((B, cap), np.uint8)
        lens = np.zeros(B, np.int64)
        _lib.check_host(_lib.host().pcgc_range_encode_lohi_batch(
            host_lohi.data_ptr(), B, seg, self._range_coder_precision, _lib.nptr(out), cap, _lib.nptr(lens),
            n_threads or _lib.host_threads()), "pcgc_range_encode_lohi_batch")
        strings = [out[i, :lens[i]].tobytes() for i in range(B)]
        return strings, mn.astype(np.int32), mx.astype(np.int32)

    def compress(self, inputs, loc, scale):
        """Reference semantics: ONE string for everything passed in, one (min_v, max_v)."""
        x = self._dev(inputs)
        s, mn, mx = self.compress_cubes(x.reshape((1,) + tuple(x.shape)), self._dev(loc).reshape((1,) + tuple(x.shape)),
                                        self._dev(scale).reshape((1,) + tuple(x.shape)), n_threads=1)
        return s[0], int(mn[0]), int(mx[0])

    # -- decode ----------------------------------------------------------
    def decompress_cubes(self, strings, locs, scales, min_vs, max_vs, datashape, n_threads=None):
        locs, scales = self._dev(locs), self._dev(scales)
        B = len(strings)
        datashape = tuple(int(s) for s in datashape)
        per_cube = int(np.prod(datashape))
        out_shape = (B,) + datashape[1:] if datashape[0] == 1 else (B,) + datashape
        if B == 0:
            return torch.empty(out_shape, dtype=torch.float32, device=locs.device)
        assert locs.numel() == B * per_cube and scales.numel() == B * per_cube
        mn = np.ascontiguousarray(min_vs, np.int32).reshape(B)
        mx = np.ascontiguousarray(max_vs, np.int32).reshape(B)
        ncols = self._check_range(mn, mx)
        rows = B * per_cube
        mn_d, mx_d = torch.from_numpy(mn).to(locs.device), torch.from_numpy(mx).to(locs.device)
        cdf = torch.empty((rows, ncols), dtype=torch.int16, device=locs.device)        # uint16 payload
        _lib.check(_lib.hip().pcgc_laplace_cdf(_lib.dptr(locs), _lib.dptr(scales), _lib.dptr(mn_d), _lib.dptr(mx_d), rows,
                                               per_cube, ncols, self._likelihood_bound, None, _lib.dptr(cdf), None,
                                               _lib.stream()), "pcgc_laplace_cdf")
        host_cdf = self._pin("cdf", (rows, ncols), torch.int16)
        host_cdf.copy_(cdf, non_blocking=True)
        lens = np.array([len(s) for s in strings], np.int64)
        offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        blob = np.frombuffer(b"".join(bytes(s) for s in strings) + b"\0", np.uint8)
        n_sym = (mx - mn + 1).astype(np.int32)
        sym = self._pin("sym", (rows,), torch.int16)
        torch.cuda.current_stream().synchronize()
        _lib.check_host(_lib.host().pcgc_range_decode_u16_batch(
            _lib.nptr(blob), _lib.nptr(offsets), _lib.nptr(lens), B, per_cube, host_cdf.data_ptr(), ncols,
            _lib.nptr(n_sym), self._range_coder_precision, sym.data_ptr(), n_threads or _lib.host_threads()),
            "pcgc_range_decode_u16_batch")
        sym_d = sym.to(locs.device, non_blocking=True).to(torch.float32).reshape(B, per_cube)
        y = sym_d + mn_d.to(torch.float32).reshape(B, 1)
        return y.reshape(out_shape)

    def decompress(self, strings, loc, scale, min_v, max_v, datashape):
        loc = self._dev(loc)
        datashape = tuple(int(s) for s in datashape)
        y = self.decompress_cubes([strings], loc.reshape((1,) + tuple(loc.shape)),
                                  self._dev(scale).reshape((1,) + tuple(loc.shape)), [min_v], [max_v],
                                  (1, int(np.prod(datashape))), n_threads=1)
        return y.reshape(datashape)
