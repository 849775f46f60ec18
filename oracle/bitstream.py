"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

Restatement of the reference's on-disk container for the hyperprior mode:

  dataprocess/inout_bitstream.py:75-141   write_binary_files_hyper
  dataprocess/inout_bitstream.py:144-198  read_binary_files_hyper
  dataprocess/inout_bitstream.py:10-70    write/read_binary_files_factorized

PINNED: tools/make_golden.py runs the reference writer (numpy-only, importable
here; tmc3 runs from /root/reference) on hand-made inputs and commits the
produced bytes in tests/golden/bitstream_hyper.npz;
tests/test_oracle_bitstream.py checks these functions against them byte for
byte.  `.cubepos` is produced by the prebuilt `myutils/tmc3` (G-PCC v6) in the
reference; that binary cannot travel, so the cube-position stream is handled by
the product's own codec and only its decoded positions are compared.
All integers little-endian.
"""
import numpy as np


def pack_strings_head(y_strings, y_min_vs, y_max_vs, y_shape):
    """inout_bitstream.py:92-105."""
    out = bytearray()
    out += np.array(len(y_strings), dtype=np.int16).tobytes()
    mm = np.asarray(y_max_vs) * 16 - np.asarray(y_min_vs)
    out += np.array(mm, dtype=np.uint8).tobytes()
    for s in y_strings:
        l = len(s)
        if l <= 255:
            out += np.array(l, dtype=np.uint8).tobytes()
        else:
            out += np.array(0, dtype=np.uint8).tobytes()
            out += np.array(l, dtype=np.int16).tobytes()
    out += np.array(y_shape, dtype=np.int16).tobytes()
    return bytes(out)


def pack_strings(y_strings):
    return b"".join(bytes(s) for s in y_strings)


def pack_strings_hyper(z_string, z_min_v, z_max_v, z_shape):
    """inout_bitstream.py:111-114 (also the factorized `.strings`, 27-30)."""
    return (np.array(z_shape, dtype=np.int16).tobytes()
            + np.array((z_min_v, z_max_v), dtype=np.int8).tobytes() + bytes(z_string))


def pack_pointnums(points_numbers):
    return np.array(points_numbers, dtype=np.uint16).tobytes()


def unpack_strings_head(buf):
    """inout_bitstream.py:159-177. Returns (y_min_vs, y_max_vs, lens, y_shape)."""
    n = int(np.frombuffer(buf[:2], dtype=np.int16)[0])
    mm = np.frombuffer(buf[2:2 + n], dtype=np.uint8).astype("int32")
    y_max_vs = mm // 16
    y_min_vs = -(mm % 16)
    pos = 2 + n
    lens = []
    for _ in range(n):
        l = buf[pos]
        pos += 1
        if l == 0:
            l = int(np.frombuffer(buf[pos:pos + 2], dtype=np.int16)[0])
            pos += 2
        lens.append(int(l))
    y_shape = np.frombuffer(buf[pos:pos + 10], dtype=np.int16)
    return y_min_vs, y_max_vs, np.array(lens, dtype=np.int32), y_shape


def unpack_strings(buf, lens):
    out, pos = [], 0
    for l in lens:
        out.append(bytes(buf[pos:pos + int(l)]))
        pos += int(l)
    return out


def unpack_strings_hyper(buf):
    z_shape = np.frombuffer(buf[:10], dtype=np.int16)
    z_min_v, z_max_v = np.frombuffer(buf[10:12], dtype=np.int8)
    return bytes(buf[12:]), int(z_min_v), int(z_max_v), z_shape


def unpack_pointnums(buf):
    return np.frombuffer(buf, dtype=np.uint16)
