"""SymmetricConditional (Laplace prior conditioned on loc / scale) — operator surface of
the reference's models/conditional_entropy_model.py (class SymmetricConditional, 8-201).

  __call__(inputs, loc, scale, training) -> (values, likelihood)      :71-93
  compress(inputs, loc, scale)           -> (string, min_v, max_v)     :126-163
  decompress(strings, loc, scale, min_v, max_v, datashape)            :165-201

plus the batched forms compress_hyper / decompress_hyper use instead of the reference's
per-cube tf.map_fn (transform.py:157-168, 238-248):

  compress_cubes(ys, locs, scales)  -> (list of strings, min_vs, max_vs)   one string per cube
  decompress_cubes(strings, locs, scales, min_vs, max_vs, datashape)
  decompress_slices(...)            -> generator of (lo, hi, y_hat[lo:hi]) as soon as each slice is decoded

Device side (libpcgc_hip.so): rounding + per-cube min/max, the Laplace pmf table and its
16-bit quantised CDF for every (voxel, channel) row.  Host side (libpcgc_host.so): the
sequential range coder, one thread per cube stream.  The batch is cut into a few slices of cubes:
CDF kernels and device->host copies run ahead (the decoder's on an entropy stream of their own), and the
host codes slice k while the GPU still works on slice k+1 (and, when decoding, while it already
synthesises slice k-1).
"""
import numpy as np
import torch

from .. import _lib

_MAX_SYMBOLS = 32
_SLICES = int(__import__("os").environ.get("PCGC_SLICES", "2"))   # measured on the 205-cube batch: 1 -> 2 slices +2 %, 4 slower


_SLICE_ALIGN = int(__import__("os").environ.get("PCGC_SLICE_ALIGN", "8"))


def _slices(B, n):
    """n nearly equal slices whose boundaries are multiples of 8 cubes = the launch size of the 64^3 stage (a 103-cube
    group as 52 + 51 ends both synthesis passes with a half-filled 4- / 3-cube launch; 56 + 47 ends them with 8 and 7)."""
    n = max(1, min(n, B // 32))          # a slice keeps >= 32 cubes (the 32^3 / 16^3 stages want large chunks)
    base, rem = divmod(B, n)
    out, lo = [], 0
    for i in range(n):
        hi = lo + base + (1 if i < rem else 0)
        if i + 1 < n and _SLICE_ALIGN > 1:
            hi = min(B, -(-hi // _SLICE_ALIGN) * _SLICE_ALIGN)
        hi = B if i + 1 == n else hi
        out.append((lo, hi))
        lo = hi
    return [s_ for s_ in out if s_[1] > s_[0]]


def decode_slices(B, first=None, n=None, row_bytes=0, tail=0):
    """Slice boundaries of one decoder pipeline: a SHORT first slice (24 cubes = three launches of the 64^3 stage) followed
    by ONE slice with the rest of the pipeline's cubes (several of about 100 cubes for a large cloud).  Nothing runs on the GPU until the first slice's symbols are decoded — its share of the z stream,
    its hyper decoder, CDF rows, their copy and its strings all sit on the critical path — while the rest hide behind the
    synthesis of their predecessors.  Measured (PCGC_FIRST_SLICE sweeps): within noise while the entropy
    stream of a pipeline shared a hardware queue with its synthesis stream; with 8 queues 24 cubes give 47.9-48.2 ms per
    205-cube round trip against 48.6-49.4 ms for equal slices (16 the same, 32 less); wide CDF rows (12+ symbols: 125 MB
    per 50 cubes on their way to the host) gained from it before.  The rest as one slice instead of two (79 cubes per
    synthesis call instead of 40 + 39: larger launches at 32^3 / 16^3): 49.4 against 49.8 ms, better in five of six
    interleaved pairs of 60-step runs.  `row_bytes` is kept for callers that pass it.  tail > 0 (a caller that consumes
    the slices as they finish, process.StreamedPostprocess): the rest ends with a slice of about `tail` cubes, so that
    little is left to do once the GPU is through."""
    first = (_FIRST_SLICE if _FIRST_SLICE >= 0 else 24) if first is None else first
    # a pipeline of many hundred cubes (a vox12 cloud: thousands of cubes) keeps slices of about 100 cubes: what the GPU waits
    # for at the start — the first slice's share of the z stream, its rows, its strings — does not grow with the cloud
    n = max(_DEC_SLICES, B // 100) if n is None else n
    if first <= 0 or B < first + 32:
        return _slices(B, n)
    rest = B - first
    if tail > 0 and n == 1 and rest >= 2 * tail + 16:
        cut = ((B - tail) // 8) * 8                      # on a multiple of 8 like every other boundary (the 64^3 launches)
        return [(0, first), (first, cut), (cut, B)]
    return [(0, first)] + [(first + lo, first + hi) for lo, hi in _slices(rest, n)]


_DEC_SLICES = int(__import__("os").environ.get("PCGC_DEC_SLICES", "1"))      # decoder slices after the first one (below 200 cubes)
_FIRST_SLICE = int(__import__("os").environ.get("PCGC_FIRST_SLICE", "-1"))     # -1: the default of decode_slices; 0: none


class SymmetricConditional(object):
    def __init__(self, likelihood_bound=1e-9, range_coder_precision=16):
        self._likelihood_bound = float(likelihood_bound)
        self._range_coder_precision = int(range_coder_precision)
        if self._range_coder_precision != 16:
            raise NotImplementedError("the device CDF kernel emits 16-bit CDFs (the reference's only setting)")
        self._pinned = {}
        self._guards = {}

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _dev(t):
        dev = _lib.require_gpu()
        if not torch.is_tensor(t):
            t = torch.from_numpy(np.ascontiguousarray(t, np.float32))
        return t.to(dev, torch.float32).contiguous()

    def _side_stream(self, role, cur):
        """the `role` stream that belongs to the stream `cur` (one set per pipeline, shared by every codec of the process)"""
        return _lib.side_stream(role, cur)

    def _guard(self, what, cur):
        """Host-side wait for the last asynchronous upload that read this pipeline's pinned staging buffers `what`, before the
        host overwrites them (normally long finished; matters when calls follow each other without a synchronisation)."""
        ev = self._guards.pop((what, int(cur.cuda_stream)), None)
        if ev is not None:
            ev.synchronize()

    def _pin(self, key, shape, dtype):
        key = (key, int(torch.cuda.current_stream().cuda_stream))      # one set of staging buffers per stream / pipeline
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
        # the container byte y_max*16 - y_min needs min <= 0 <= max (inout_bitstream.py:95-96, 163-164): a cube whose
        # symbols are all positive (or all negative) is coded over a support that includes 0 — the range travels in the
        # header, so the decoder builds the same CDF (the reference writes a corrupt header for such a cube)
        mn, mx = np.minimum(mn, 0), np.maximum(mx, 0)
        same = mx == mn
        if same.any():
            mx = np.where(same & (mx < 15), mx + 1, mx)
            mn = np.where(same & (mx == mn), mn - 1, mn)
        return mn.astype(np.int32), mx.astype(np.int32)

    def start_ranges(self, ys):
        """First half of compress_cubes, to be queued as soon as the latents exist: rounding + per-cube min / max and their
        copy to the host.  The hyper encoder / decoder launches that follow hide the round trip; compress_cubes(...,
        ranges=this) then finds the ranges on the host instead of stalling on them."""
        ys = self._dev(ys)
        B = int(ys.shape[0])
        if B == 0:
            return None
        y_hat, mn_d, mx_d = self.quantize_minmax(ys, B)
        host_mm = self._pin("minmax", (2, B), torch.int32)
        host_mm[0].copy_(mn_d, non_blocking=True)
        host_mm[1].copy_(mx_d, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return y_hat, mn_d, mx_d, host_mm, ev

    def compress_cubes(self, ys, locs, scales, n_threads=None, n_slices=_SLICES, ranges=None):
        ys, locs, scales = self._dev(ys), self._dev(locs), self._dev(scales)
        B = int(ys.shape[0])
        if B == 0:
            return [], np.zeros(0, np.int32), np.zeros(0, np.int32)
        rows = ys.numel()
        seg = rows // B
        lib, host = _lib.hip(), _lib.host()
        y_hat, mn_d, mx_d, host_mm, ev = ranges if ranges is not None else self.start_ranges(ys)
        ev.synchronize()
        _lib.mark("enc ranges on the host")
        mn0, mx0 = host_mm[0].numpy().copy(), host_mm[1].numpy().copy()
        mn, mx = self._widen(mn0, mx0)
        if not (np.array_equal(mn, mn0) and np.array_equal(mx, mx0)):
            mn_d, mx_d = torch.from_numpy(mn).to(ys.device), torch.from_numpy(mx).to(ys.device)
        ncols = self._check_range(mn, mx)
        lohi = torch.empty(rows, dtype=torch.int32, device=ys.device)
        host_lohi = self._pin("lohi", (rows,), torch.int32)
        yf, lf, sf = y_hat.reshape(-1), locs.reshape(-1), scales.reshape(-1)
        events = []
        for lo, hi in _slices(B, n_slices):             # queue every slice's kernel + copy, then code as they land
            a, b = lo * seg, hi * seg
            _lib.check(lib.pcgc_laplace_cdf(_lib.dptr(lf[a:b]), _lib.dptr(sf[a:b]), _lib.dptr(mn_d[lo:hi]), _lib.dptr(mx_d[lo:hi]),
                                            b - a, seg, ncols, self._likelihood_bound, _lib.dptr(yf[a:b]), None,
                                            _lib.dptr(lohi[a:b]), _lib.stream()), "pcgc_laplace_cdf")
            host_lohi[a:b].copy_(lohi[a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events.append((lo, hi, ev))
        cap = seg * 2 + 1024
        # reused across calls: a fresh 13 MB array costs its page faults inside the first coder batch (1.6 ms against 0.3 ms)
        out = self._pin("enc_out", (B, cap), torch.uint8).numpy()
        lens = np.zeros(B, np.int64)
        nt = n_threads or _lib.host_threads()
        _lib.mark("enc cdf slices queued")
        for lo, hi, ev in events:
            ev.synchronize()
            _lib.mark("enc slice [%d:%d] on the host" % (lo, hi))
            _lib.check_host(host.pcgc_range_encode_lohi_batch(host_lohi[lo * seg:].data_ptr(), hi - lo, seg,
                                                              self._range_coder_precision, _lib.nptr(out[lo:hi]), cap,
                                                              _lib.nptr(lens[lo:hi]), nt), "pcgc_range_encode_lohi_batch")
        strings = [out[i, :lens[i]].tobytes() for i in range(B)]
        return strings, mn.astype(np.int32), mx.astype(np.int32)

    def compress(self, inputs, loc, scale):
        """Reference semantics: ONE string for everything passed in, one (min_v, max_v)."""
        x = self._dev(inputs)
        s, mn, mx = self.compress_cubes(x.reshape((1,) + tuple(x.shape)), self._dev(loc).reshape((1,) + tuple(x.shape)),
                                        self._dev(scale).reshape((1,) + tuple(x.shape)), n_threads=1)
        return s[0], int(mn[0]), int(mx[0])

    # -- decode ----------------------------------------------------------
    def decompress_slices(self, strings, locs, scales, min_vs, max_vs, datashape, n_threads=None, n_slices=_SLICES, slices=None):
        """Yields (lo, hi, y_hat[lo:hi]) — float32 device tensors shaped like the encoder's latents — slice by
        slice, so the caller can start the synthesis of a slice while the host decodes the next one.
        `locs` may be a callable hd(lo, hi) -> (locs, scales) of cubes lo..hi-1 (scales is then ignored): the hyper decoder
        runs per slice, so the first slice needs only ITS share of the sequential z stream (transform.decompress_hyper).
        `slices`: explicit [(lo, hi)] boundaries instead of n_slices nearly equal ones.  Hyper decoder, CDF kernel and row copy
        of a slice run on the pipeline's entropy stream, symbol uploads on its upload stream (see below); slice k + 1's are
        queued right after slice k has been handed to the caller.  With the tables given (not lazy) every slice's kernel and
        copy are queued up front."""
        lazy = callable(locs)
        if not lazy:
            locs, scales = self._dev(locs), self._dev(scales)
        B = len(strings)
        datashape = tuple(int(s) for s in datashape)
        per_cube = int(np.prod(datashape))
        cube_shape = datashape[1:] if datashape[0] == 1 else datashape
        if B == 0:
            return
        dev = _lib.require_gpu()
        assert lazy or (locs.numel() == B * per_cube and scales.numel() == B * per_cube)
        lib, host = _lib.hip(), _lib.host()
        mn = np.ascontiguousarray(min_vs, np.int32).reshape(B)
        mx = np.ascontiguousarray(max_vs, np.int32).reshape(B)
        ncols = self._check_range(mn, mx)
        rows = B * per_cube
        # Three streams per decoder pipeline.  The ENTROPY stream carries what produces CDF rows (hyper decoder, CDF kernel,
        # their 0.2 us-per-row copy to the host); the UPLOAD stream carries decoded symbols to the device and turns them into
        # latents; the caller's stream only waits for a slice's latents and runs the caller's synthesis.  The host decoder
        # sits between the first two, nothing on the device orders them: the 40 MB row copy of slice k + 1 never stands in
        # front of the synthesis of slice k, and an upload never waits behind the synthesis of slice k - 1 (so the pinned
        # staging buffers are consumed at once, whatever the caller queues next).  Order of issue still matters: both copy
        # directions share a queue here (a symbol upload issued behind a row download waits for it: 0.8 ms measured), hence
        # "upload slice k, THEN queue slice k + 1".
        cur = torch.cuda.current_stream()
        es, us = self._side_stream("entropy", cur), self._side_stream("upload", cur)
        start = torch.cuda.Event()
        start.record(cur)
        es.wait_event(start)
        us.wait_event(start)
        todo = list(slices) if slices else _slices(B, n_slices)
        first = None
        if lazy:
            # the first slice's hyper decoder goes to the device before anything else: the buffers set up below (pinned
            # staging, the row tensor, the range upload) are host work the GPU does not have to wait for
            with torch.cuda.stream(es):
                first = locs(*todo[0])
        self._guard("dec_uploads", cur)                              # the previous call's uploads have read mm_up / sym
        mm_host = self._pin("mm_up", (3, B), torch.float32)          # one upload: min, max (as int32 bits) and min as float
        mm_host[0:2].view(torch.int32).copy_(torch.from_numpy(np.stack([mn, mx])))
        mm_host[2].copy_(torch.from_numpy(mn.astype(np.float32)))
        with torch.cuda.stream(es):
            mm_d = mm_host[0:2].to(dev, non_blocking=True)
            mn_d, mx_d = mm_d[0].view(torch.int32), mm_d[1].view(torch.int32)
            cdf = torch.empty((rows, ncols), dtype=torch.int16, device=dev)        # uint16 payload
        with torch.cuda.stream(us):
            mn_f = mm_host[2].to(dev, non_blocking=True)
        host_cdf = self._pin("cdf", (rows, ncols), torch.int16)
        if not lazy:
            lf, sf = locs.reshape(-1), scales.reshape(-1)

        def queue(lo, hi):
            nonlocal first
            a, b = lo * per_cube, hi * per_cube
            with torch.cuda.stream(es):
                if lazy:
                    l_, s_ = first if first is not None else locs(lo, hi)
                    first = None
                    l_, s_ = self._dev(l_).reshape(-1), self._dev(s_).reshape(-1)
                    assert l_.numel() == b - a and s_.numel() == b - a
                else:
                    l_, s_ = lf[a:b], sf[a:b]
                _lib.check(lib.pcgc_laplace_cdf(_lib.dptr(l_), _lib.dptr(s_), _lib.dptr(mn_d[lo:hi]), _lib.dptr(mx_d[lo:hi]),
                                                b - a, per_cube, ncols, self._likelihood_bound, None, _lib.dptr(cdf[a:b]), None,
                                                _lib.stream()), "pcgc_laplace_cdf")
                host_cdf[a:b].copy_(cdf[a:b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            return lo, hi, ev
        _lib.mark("dec set-up done")
        if lazy:
            events = [queue(*todo[0])]
        else:
            done = torch.cuda.Event()
            done.record(cur)
            es.wait_event(done)                          # locs / scales were produced on the caller's stream
            events = [queue(lo, hi) for lo, hi in todo]  # everything is known: queue every slice's kernel + copy up front
        # what only the host decoder needs is prepared while the device makes the first slice's rows
        lens = np.array([len(s) for s in strings], np.int64)
        offsets = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        blob = np.frombuffer(b"".join(bytes(s) for s in strings) + b"\0", np.uint8)
        n_sym = (mx - mn + 1).astype(np.int32)
        sym = self._pin("sym", (rows,), torch.int16)
        nt = n_threads or _lib.host_threads()
        try:
            for k in range(len(todo)):
                lo, hi, ev = events[k]
                ev.synchronize()
                a = lo * per_cube
                _lib.check_host(host.pcgc_range_decode_u16_batch(
                    _lib.nptr(blob), _lib.nptr(offsets[lo:hi]), _lib.nptr(lens[lo:hi]), hi - lo, per_cube,
                    host_cdf[a:].data_ptr(), ncols, _lib.nptr(n_sym[lo:hi]), self._range_coder_precision, sym[a:].data_ptr(), nt),
                    "pcgc_range_decode_u16_batch")
                with torch.cuda.stream(us):
                    s_d = sym[a:hi * per_cube].to(dev, non_blocking=True)
                    y = torch.empty((hi - lo,) + tuple(cube_shape), dtype=torch.float32, device=dev)
                    _lib.check(lib.pcgc_symbols_to_values_seg(_lib.dptr(s_d), _lib.dptr(mn_f[lo:hi]), _lib.dptr(y), s_d.numel(), per_cube,
                                                              _lib.stream()), "pcgc_symbols_to_values_seg")
                    up = torch.cuda.Event()
                    up.record(us)
                y.record_stream(cur)                     # made on the upload stream, consumed on the caller's
                cur.wait_event(up)
                self._guards[("dec_uploads", int(cur.cuda_stream))] = up
                _lib.mark("dec slice [%d:%d] symbols queued" % (lo, hi))
                # the caller gets slice k (and launches its synthesis) BEFORE slice k + 1's hyper decoder, CDF rows and their
                # copy are queued: they go to the entropy stream, so nothing of it needs to precede the synthesis on the
                # device, and the symbol upload above is already ahead of the row download in the copy queue.  The host
                # decodes slice k + 1 while the device synthesises slice k.
                yield lo, hi, y
                if lazy and k + 1 < len(todo):
                    events.append(queue(*todo[k + 1]))
        finally:
            for side in (es, us):
                end = torch.cuda.Event()
                end.record(side)
                cur.wait_event(end)                      # the caller's stream outlives everything this call queued

    def decompress_cubes(self, strings, locs, scales, min_vs, max_vs, datashape, n_threads=None):
        locs = self._dev(locs)
        parts = [y for _, _, y in self.decompress_slices(strings, locs, scales, min_vs, max_vs, datashape, n_threads)]
        if not parts:
            datashape = tuple(int(s) for s in datashape)
            return torch.empty((0,) + (datashape[1:] if datashape[0] == 1 else datashape), dtype=torch.float32,
                               device=locs.device)
        return torch.cat(parts, 0)

    def decompress(self, strings, loc, scale, min_v, max_v, datashape):
        loc = self._dev(loc)
        datashape = tuple(int(s) for s in datashape)
        y = self.decompress_cubes([strings], loc.reshape((1,) + tuple(loc.shape)),
                                  self._dev(scale).reshape((1,) + tuple(loc.shape)), [min_v], [max_v],
                                  (1, int(np.prod(datashape))), n_threads=1)
        return y.reshape(datashape)
