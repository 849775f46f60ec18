"""On-disk container — same entry points and byte layout as the reference's
dataprocess/inout_bitstream.py (factorized 10-70, hyper 75-198):

  <name>.strings        concatenated per-cube y strings
  <name>.strings_head   int16 B | uint8[B] (y_max*16 - y_min) | per cube length (1 byte, or 0x00 + int16
                        when > 255) | int16[5] y_shape
  <name>.strings_hyper  int16[5] z_shape | int8 z_min | int8 z_max | z string
  <name>.pointnums      uint16[B]
  <name>.cubepos        lossless cube positions

The first four files are byte-identical to the reference writer (tests/golden/bitstream_hyper.npz).
`.cubepos` differs: the reference shells out to the prebuilt MPEG `tmc3` (myutils/gpcc_wrapper.py:5-42),
which cannot ship; here the uint8 positions go through a small octree occupancy coder on the host range
coder (libpcgc_host.so).  The reader fixes two latent bugs of the reference reader instead of copying them
(mixed <=255 / >255 lengths, inout_bitstream.py:168-174); the writer refuses what the format cannot hold
(|min|,|max| > 15, zero-length strings, lengths > 32767).
"""
import os
from math import comb

import numpy as np

from .. import coder_ops


# ------------------------------------------------------------------ cube positions
def _occupancy_cdf():
    mass = {1: 0.30, 2: 0.25, 3: 0.16, 4: 0.11, 5: 0.07, 6: 0.05, 7: 0.03, 8: 0.03}
    v = np.zeros(256, np.int64)
    for b in range(1, 256):
        pc = bin(b).count("1")
        v[b] = max(1, int(mass[pc] / comb(8, pc) * 65536))
    v[0] = 1                                     # never coded; keeps every interval non-empty
    v[255] += 65536 - v.sum()                    # integer-only normalisation
    return np.concatenate([[0], np.cumsum(v)]).astype(np.int32).reshape(1, 1, 257)


_OCC_CDF = _occupancy_cdf()


def encode_cube_positions(cube_positions):
    """uint8 [B,3] positions (inout_bitstream.py:119 casts to uint8 too) -> bytes."""
    p = np.unique(np.asarray(cube_positions).astype(np.uint8).astype(np.int64).reshape(-1, 3), axis=0)
    nb = max(1, int(p.max()).bit_length()) if len(p) else 1
    # plain Python tuples and lists: a few hundred positions, four to eight levels.  (One numpy selection per child per node
    # took 3 ms for 205 cubes — with the interpreter lock held, while the encoder's pipeline threads were trying to start.)
    symbols = []
    nodes = [[tuple(int(v) for v in row) for row in p]]       # breadth-first, parent-major, children in ascending order
    for level in range(nb):
        shift = nb - 1 - level
        nxt = []
        for pts in nodes:
            kids = [[] for _ in range(8)]
            for q in pts:
                kids[((q[0] >> shift) & 1) * 4 + ((q[1] >> shift) & 1) * 2 + ((q[2] >> shift) & 1)].append(q)
            occ = 0
            for c in range(8):
                if kids[c]:
                    occ |= 1 << c
                    nxt.append(kids[c])
            symbols.append(occ)
        nodes = nxt
    sym = np.array(symbols, np.int16).reshape(-1, 1)
    body = coder_ops.range_encode(sym, _OCC_CDF) if len(sym) else b""
    return bytes([nb]) + np.array(len(sym), np.uint16).tobytes() + body


def decode_cube_positions(buf):
    nb = buf[0]
    nsym = int(np.frombuffer(buf[1:3], np.uint16)[0])
    sym = coder_ops.range_decode(buf[3:], (nsym, 1), _OCC_CDF).reshape(-1) if nsym else np.zeros(0, np.int16)
    nodes = [(0, 0, 0)]
    i = 0
    for _ in range(nb):
        nxt = []
        for key in nodes:
            occ = int(sym[i])
            i += 1
            for c in range(8):
                if occ & (1 << c):
                    nxt.append((key[0] * 2 + (c >> 2), key[1] * 2 + ((c >> 1) & 1), key[2] * 2 + (c & 1)))
        nodes = nxt
    return np.array(nodes, np.int32).reshape(-1, 3) if nsym else np.zeros((0, 3), np.int32)


# ------------------------------------------------------------------ hyper mode
def _paths(filename, rootdir):
    return {e: os.path.join(rootdir, filename + "." + e)
            for e in ("strings", "strings_head", "strings_hyper", "pointnums", "cubepos")}


def pack_strings_head(y_strings, y_min_vs, y_max_vs, y_shape):
    y_min_vs, y_max_vs = np.asarray(y_min_vs, np.int64), np.asarray(y_max_vs, np.int64)
    if len(y_strings) > 32767:
        raise ValueError("more than 32767 cubes do not fit the int16 count of .strings_head")
    if np.any(y_max_vs > 15) or np.any(y_max_vs < 0) or np.any(y_min_vs < -15) or np.any(y_min_vs > 0):
        raise ValueError("the container packs y_max*16 - y_min into one byte: needs 0 <= max <= 15, -15 <= min <= 0")
    out = bytearray(np.array(len(y_strings), dtype=np.int16).tobytes())
    out += np.array(y_max_vs * 16 - y_min_vs, dtype=np.uint8).tobytes()
    for s in y_strings:
        l = len(s)
        if l == 0 or l > 32767:
            raise ValueError("string length %d cannot be represented (0 is the escape byte, int16 caps at 32767)" % l)
        if l <= 255:
            out += np.array(l, dtype=np.uint8).tobytes()
        else:
            out += np.array(0, dtype=np.uint8).tobytes() + np.array(l, dtype=np.int16).tobytes()
    out += np.array(y_shape, dtype=np.int16).tobytes()
    return bytes(out)


def unpack_strings_head(buf):
    n = int(np.frombuffer(buf[:2], dtype=np.int16)[0])
    mm = np.frombuffer(buf[2:2 + n], dtype=np.uint8).astype("int32")
    y_max_vs, y_min_vs = mm // 16, -(mm % 16)
    pos, lens = 2 + n, []
    for _ in range(n):
        l = buf[pos]
        pos += 1
        if l == 0:
            l = int(np.frombuffer(buf[pos:pos + 2], dtype=np.int16)[0])
            pos += 2
        lens.append(int(l))
    return y_min_vs, y_max_vs, np.array(lens, np.int32), np.frombuffer(buf[pos:pos + 10], dtype=np.int16)


def write_binary_files_hyper(filename, y_strings, z_strings, points_numbers, cube_positions, y_min_vs, y_max_vs,
                             y_shape, z_min_v, z_max_v, z_shape, rootdir='./', verbose=True, cubepos=None):
    """cubepos: encode_cube_positions(cube_positions) when the caller already has it (test.py codes the positions on a
    helper thread while the GPU encodes the cubes)."""
    os.makedirs(rootdir, exist_ok=True)
    p = _paths(filename, rootdir)
    blobs = {
        "strings_head": pack_strings_head(y_strings, y_min_vs, y_max_vs, y_shape),
        "strings": b"".join(bytes(s) for s in y_strings),
        "strings_hyper": np.array(z_shape, dtype=np.int16).tobytes()
        + np.array((z_min_v, z_max_v), dtype=np.int8).tobytes() + bytes(z_strings),
        "pointnums": np.array(points_numbers, dtype=np.uint16).tobytes(),
        "cubepos": encode_cube_positions(cube_positions) if cubepos is None else cubepos,
    }
    for k, v in blobs.items():
        with open(p[k], "wb") as f:
            f.write(v)
    sizes = tuple(len(blobs[k]) for k in ("strings", "strings_head", "strings_hyper", "pointnums", "cubepos"))
    if verbose:
        print('===== Write binary files =====')
        print('Total file size (Bytes): {}'.format(sum(sizes)))
        for name, s in zip(('Strings', 'Strings head', 'Strings hyper', 'Numbers of points', 'Positions of cubes'), sizes):
            print('{} (Bytes): {}'.format(name, s))
    return sizes


def read_binary_files_hyper(filename, rootdir='./'):
    p = _paths(filename, rootdir)
    blobs = {k: open(v, "rb").read() for k, v in p.items()}
    y_min_vs, y_max_vs, lens, y_shape = unpack_strings_head(blobs["strings_head"])
    y_strings, pos = [], 0
    for l in lens:
        y_strings.append(blobs["strings"][pos:pos + int(l)])
        pos += int(l)
    h = blobs["strings_hyper"]
    z_shape = np.frombuffer(h[:10], dtype=np.int16)
    z_min_v, z_max_v = (int(v) for v in np.frombuffer(h[10:12], dtype=np.int8))
    points_numbers = np.frombuffer(blobs["pointnums"], dtype=np.uint16)
    cube_positions = decode_cube_positions(blobs["cubepos"])
    return y_strings, h[12:], points_numbers, cube_positions, y_min_vs, y_max_vs, y_shape, z_min_v, z_max_v, z_shape


# ------------------------------------------------------------------ factorized mode
def write_binary_files_factorized(filename, strings, points_numbers, cube_positions, min_v, max_v, shape, rootdir='./',
                                  verbose=True):
    os.makedirs(rootdir, exist_ok=True)
    p = _paths(filename, rootdir)
    blobs = {"strings": np.array(shape, dtype=np.int16).tobytes() + np.array((min_v, max_v), dtype=np.int8).tobytes()
             + bytes(strings),
             "pointnums": np.array(points_numbers, dtype=np.uint16).tobytes(),
             "cubepos": encode_cube_positions(cube_positions)}
    for k, v in blobs.items():
        with open(p[k], "wb") as f:
            f.write(v)
    sizes = tuple(len(blobs[k]) for k in ("strings", "pointnums", "cubepos"))
    if verbose:
        print('===== Write binary files =====')
        print('Total file size (Bytes): {}'.format(sum(sizes)))
    return sizes


def read_binary_files_factorized(filename, rootdir='./'):
    p = _paths(filename, rootdir)
    s = open(p["strings"], "rb").read()
    shape = np.frombuffer(s[:10], dtype=np.int16)
    min_v, max_v = (int(v) for v in np.frombuffer(s[10:12], dtype=np.int8))
    points_numbers = np.frombuffer(open(p["pointnums"], "rb").read(), dtype=np.uint16)
    cube_positions = decode_cube_positions(open(p["cubepos"], "rb").read())
    return s[12:], points_numbers, cube_positions, min_v, max_v, shape
