"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

ctypes front-end for oracle/coder.c (plain-C restatement of TF 1.13
contrib/coder, call sites entropy_model.py:218,258,298 and
conditional_entropy_model.py:122,161,195) with the same three entry points the
reference calls through `coder_ops`, plus an independent pure-Python
restatement (`py_*`, big-int arithmetic, small cases only) that
tests/test_oracle_coder.py plays against the C code.

*** PARITY UNPINNED *** (TensorFlow is not installable here and the reference
has no golden vectors for these ops; see coder.c header).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile coder.c -> _build/liboracle_coder.so (gcc). Building the checker
    is not using it; __graft_entry__.build() calls this."""
    so = os.path.join(_HERE, "_build", "liboracle_coder.so")
    src = os.path.join(_HERE, "coder.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.oracle_pmf_to_quantized_cdf.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_void_p]
        L.oracle_range_encode.restype = ctypes.c_int64
        L.oracle_range_encode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int64]
        L.oracle_range_decode.restype = ctypes.c_int
        L.oracle_range_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p]
        _LIB = L
    return _LIB


def pmf_to_quantized_cdf(pmf, precision=16):
    """pmf float32 [..., n] -> int32 [..., n+1]."""
    pmf = np.ascontiguousarray(pmf, np.float32)
    n = pmf.shape[-1]
    rows = pmf.size // n
    cdf = np.empty((rows, n + 1), np.int32)
    rc = _lib().oracle_pmf_to_quantized_cdf(pmf.ctypes.data, rows, n, precision, cdf.ctypes.data)
    if rc:
        raise ValueError("pmf_to_quantized_cdf failed (%d)" % rc)
    return cdf.reshape(pmf.shape[:-1] + (n + 1,))


def _norm_cdf(data_shape, cdf):
    rows, cols = data_shape
    cdf = np.ascontiguousarray(cdf, np.int32)
    n = cdf.shape[-1] - 1
    lead = int(np.prod(cdf.shape[:-1]))
    if lead == rows * cols:
        bc = 0
    elif lead == cols:
        bc = 1
    else:
        raise ValueError("cdf shape %r does not match data %r" % (cdf.shape, data_shape))
    return cdf, n, bc


def range_encode(data, cdf, precision=16):
    """data int16 [rows, cols]; cdf int32 [rows|1, cols, n+1] -> bytes."""
    data = np.ascontiguousarray(data, np.int16)
    rows, cols = data.shape
    cdf, n, bc = _norm_cdf((rows, cols), cdf)
    cap = max(64, data.size * 4 + 16)
    out = np.empty(cap, np.uint8)
    ln = _lib().oracle_range_encode(data.ctypes.data, rows, cols, cdf.ctypes.data, n, bc, precision,
                                    out.ctypes.data, cap)
    if ln < 0:
        raise ValueError("range_encode: symbol outside CDF")
    assert ln <= cap
    return out[:ln].tobytes()


def range_decode(string, shape, cdf, precision=16):
    rows, cols = int(shape[0]), int(shape[1])
    cdf, n, bc = _norm_cdf((rows, cols), cdf)
    buf = np.frombuffer(bytes(string), np.uint8)
    out = np.empty((rows, cols), np.int16)
    rc = _lib().oracle_range_decode(buf.ctypes.data if buf.size else None, buf.size, rows, cols,
                                    cdf.ctypes.data, n, bc, precision, out.ctypes.data)
    if rc:
        raise ValueError("range_decode failed")
    return out


# ----------------------------------------------------------------------------
# independent pure-Python restatement (small inputs)
# ----------------------------------------------------------------------------
def py_pmf_to_quantized_cdf_row(pmf, precision=16):
    norm = 1 << precision
    pmf32 = [np.float32(m) for m in pmf]
    v = [max(1, int(np.rint(np.float32(m) * np.float32(norm)))) for m in pmf32]
    total = sum(v)
    mass = [float(m) for m in pmf32]

    def pen(i):
        return math.inf if v[i] <= 1 else mass[i] * (math.log2(v[i]) - math.log2(v[i] - 1))

    def gain(i):
        return -math.inf if v[i] < 1 else mass[i] * (math.log2(v[i] + 1) - math.log2(v[i]))

    if total > norm:
        q = sorted(range(len(v)), key=lambda i: pen(i))          # Python's sort is stable
        keys = {i: pen(i) for i in q}
        while total > norm:
            total -= 1
            h = q[0]
            assert v[h] > 1
            v[h] -= 1
            keys[h] = pen(h)
            j = 1
            while j < len(q) and not (keys[h] < keys[q[j]]):
                j += 1
            q = q[1:j] + [h] + q[j:]
    elif total < norm:
        q = sorted(range(len(v)), key=lambda i: -gain(i))
        keys = {i: gain(i) for i in q}
        while total < norm:
            total += 1
            h = q[0]
            v[h] += 1
            keys[h] = gain(h)
            j = 1
            while j < len(q) and not (keys[h] > keys[q[j]]):
                j += 1
            q = q[1:j] + [h] + q[j:]
    cdf = [0]
    for x in v:
        cdf.append(cdf[-1] + x)
    return cdf


def py_range_encode(symbols, cdfs, precision=16):
    """symbols: list of ints; cdfs: list of per-symbol CDF lists."""
    M32 = 0xFFFFFFFF
    base, size_m1, delay = 0, M32, 0
    out = bytearray()
    for s, cdf in zip(symbols, cdfs):
        lower, upper = cdf[s], cdf[s + 1]
        size = size_m1 + 1
        a = ((size * lower) >> precision) & M32
        b = (((size * upper) >> precision) - 1) & M32
        base = (base + a) & M32
        size_m1 = (b - a) & M32
        overflow = base < a
        if ((base + size_m1) & M32) < base:
            if size_m1 >> 16 == 0:
                base = (base << 16) & M32
                size_m1 = ((size_m1 << 16) | 0xFFFF) & M32
                delay += 0x20000
            continue
        if delay != 0:
            if overflow:
                out += bytes([(delay >> 8) & 0xFF, delay & 0xFF]) + b"\x00" * (delay >> 16)
            else:
                delay -= 1
                out += bytes([(delay >> 8) & 0xFF, delay & 0xFF]) + b"\xff" * (delay >> 16)
            delay = 0
        if size_m1 >> 16 == 0:
            top = base >> 16
            base = (base << 16) & M32
            size_m1 = ((size_m1 << 16) | 0xFFFF) & M32
            if base <= ((base + size_m1) & M32):
                out += bytes([(top >> 8) & 0xFF, top & 0xFF])
            else:
                delay = top + 1
    if delay != 0:
        out.append((delay >> 8) & 0xFF)
        if delay & 0xFF:
            out.append(delay & 0xFF)
    elif base != 0:
        mid = ((base - 1) >> 16) + 1
        out.append((mid >> 8) & 0xFF)
        if mid & 0xFF:
            out.append(mid & 0xFF)
    return bytes(out)


def py_range_decode(string, cdfs, precision=16):
    M32 = 0xFFFFFFFF
    data = bytes(string)
    pos = 0
    base, size_m1, value = 0, M32, 0

    def read16():
        nonlocal value, pos
        for _ in range(2):
            value = (value << 8) & M32
            if pos < len(data):
                value |= data[pos]
                pos += 1

    read16()
    read16()
    out = []
    for cdf in cdfs:
        size = size_m1 + 1
        offset = ((((value - base) & M32) + 1) << precision) - 1
        s = 0
        while s + 1 < len(cdf) and size * cdf[s + 1] <= offset:     # linear search == the binary search's answer
            s += 1
        assert s + 1 < len(cdf)
        a = ((size * cdf[s]) >> precision) & M32
        b = (((size * cdf[s + 1]) >> precision) - 1) & M32
        base = (base + a) & M32
        size_m1 = (b - a) & M32
        if size_m1 >> 16 == 0:
            base = (base << 16) & M32
            size_m1 = ((size_m1 << 16) | 0xFFFF) & M32
            read16()
        out.append(s)
    return out
