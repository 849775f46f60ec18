"""`coder_ops` surface of the reference (tensorflow.contrib.coder.python.ops.coder_ops,
imported at models/entropy_model.py:6 and models/conditional_entropy_model.py:6) on top of
libpcgc_host.so: same three entry points, same argument meaning, numpy in / numpy out.
"""
import numpy as np

from . import _lib


def pmf_to_quantized_cdf(pmf, precision=16):
    """pmf float32 [..., n] -> quantised CDF int32 [..., n+1] (entropy_model.py:218)."""
    pmf = np.ascontiguousarray(pmf, np.float32)
    n = pmf.shape[-1]
    if n < 2:
        raise ValueError("pmf_to_quantized_cdf: the last dimension must hold at least 2 symbols "
                         "(entropy_model.py:192-193 notes the same restriction)")
    rows = pmf.size // n
    cdf = np.empty((rows, n + 1), np.int32)
    _lib.check_host(_lib.host().pcgc_pmf_to_quantized_cdf(_lib.nptr(pmf), rows, n, precision, _lib.nptr(cdf)),
                    "pmf_to_quantized_cdf")
    return cdf.reshape(pmf.shape[:-1] + (n + 1,))


def _geometry(shape, cdf):
    rows, cols = int(shape[0]), int(shape[1])
    cdf = np.ascontiguousarray(cdf, np.int32)
    n = cdf.shape[-1] - 1
    lead = int(np.prod(cdf.shape[:-1]))
    if lead == rows * cols:
        bc = 0
    elif lead == cols:
        bc = 1                                      # [1, C, N+1] broadcast over rows (entropy_model.py:219)
    else:
        raise ValueError("cdf shape %r is not broadcastable to data shape %r" % (cdf.shape, (rows, cols)))
    return rows, cols, cdf, n, bc


def range_encode(data, cdf, precision=16):
    """data int16 [rows, cols], cdf int32 [rows|1, cols, n+1] -> bytes (entropy_model.py:258)."""
    data = np.ascontiguousarray(data, np.int16)
    if data.ndim != 2:
        data = data.reshape(-1, data.shape[-1])
    rows, cols, cdf, n, bc = _geometry(data.shape, cdf)
    cap = max(64, data.size // 2 + 64)
    while True:
        out = np.empty(cap, np.uint8)
        ln = np.zeros(1, np.int64)
        rc = _lib.host().pcgc_range_encode(_lib.nptr(data), rows, cols, _lib.nptr(cdf), n, bc, precision,
                                           _lib.nptr(out), cap, _lib.nptr(ln))
        if rc == -2:
            cap = int(ln[0]) + 16
            continue
        _lib.check_host(rc, "range_encode")
        return out[:int(ln[0])].tobytes()


def range_encode_values(values, offset, cdf, precision=16):
    """range_encode(values - offset, cdf) without materialising the symbols: values int8 / int16 [rows, cols] (rounded
    latents as they come off the device), offset = min_v."""
    values = np.ascontiguousarray(values)
    assert values.dtype in (np.int8, np.int16) and values.ndim == 2
    rows, cols, cdf, n, bc = _geometry(values.shape, cdf)
    cap = max(64, values.size // 2 + 64)
    while True:
        out = np.empty(cap, np.uint8)
        ln = np.zeros(1, np.int64)
        rc = _lib.host().pcgc_range_encode_values(_lib.nptr(values), values.dtype.itemsize, rows, cols, int(offset), _lib.nptr(cdf), n,
                                                  bc, precision, _lib.nptr(out), cap, _lib.nptr(ln))
        if rc == -2:
            cap = int(ln[0]) + 16
            continue
        _lib.check_host(rc, "range_encode_values")
        return out[:int(ln[0])].tobytes()


def range_decode(encoded, shape, cdf, precision=16):
    """bytes -> int16 [rows, cols] (entropy_model.py:298)."""
    rows, cols, cdf, n, bc = _geometry(shape, cdf)
    buf = np.frombuffer(bytes(encoded), np.uint8)
    out = np.empty((rows, cols), np.int16)
    _lib.check_host(_lib.host().pcgc_range_decode(_lib.nptr(buf) if buf.size else None, buf.size, rows, cols,
                                                  _lib.nptr(cdf), n, bc, precision, _lib.nptr(out)), "range_decode")
    return out


def range_decode_async(encoded, shape, cdf, precision=16):
    """range_decode on a helper thread.  Returns (out, wait): `out` is the int16 [rows, cols] array being filled in
    row order, `wait(rows_needed)` blocks until that many leading rows are final (raises on a corrupt stream)."""
    import threading
    import time
    rows, cols, cdf, n, bc = _geometry(shape, cdf)
    buf = np.frombuffer(bytes(encoded), np.uint8)
    out = np.empty((rows, cols), np.int16)
    progress = np.zeros(1, np.int64)
    box = {}

    def work():
        box["rc"] = _lib.host().pcgc_range_decode_progress(_lib.nptr(buf) if buf.size else None, buf.size, rows, cols,
                                                           _lib.nptr(cdf), n, bc, precision, _lib.nptr(out), _lib.nptr(progress))
        if box["rc"] != 0:
            box["msg"] = _lib.host().pcgc_host_last_error().decode()
    th = _lib.workers().submit(work)

    def wait(rows_needed, block=True):
        rows_needed = min(int(rows_needed), rows)
        while True:
            done = int(progress[0])
            if done < 0 or ("rc" in box and box["rc"] != 0):
                th.result()
                raise _lib.PcgcError("range_decode failed: %s" % box.get("msg", "corrupt stream"))
            if done >= rows_needed:
                if rows_needed == rows:
                    th.result()
                return True
            if not block:
                return False
            time.sleep(0.00005)
    return out, wait
