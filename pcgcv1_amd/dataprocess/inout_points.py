"""Point-cloud <-> cube handling — same function names as the reference's
dataprocess/inout_points.py, vectorised / on the device instead of per-point Python loops.

  load_ply_data 8-28, write_ply_data 30-46, load_points 50-90 (partition), save_points 92-112
  (merge), points2voxels 116-132, voxels2points 134-143, select_voxels 147-168,
  get_adaptive_thres 170-179.

Bit-exact against the reference on the golden vectors (tests/golden/partition.npz, select.npz).
Partition runs in libpcgc_host.so (`pcgc_partition`), voxelisation and the adaptive top-k
threshold in libpcgc_hip.so (`pcgc_voxelize`, `pcgc_topk_threshold`).
"""
import io
import os

import numpy as np

from .. import _lib


# ---------------------------------------------------------------------------- ply text
def load_ply_data(filename):
    """ASCII ply -> int32 [N,3].  Like the reference (inout_points.py:8-28), every line whose first three
    single-space-separated tokens parse as floats is a point (header lines fail to parse and are skipped); values
    are truncated to int32.  The text is parsed by libpcgc_host.so on a few threads (pcgc_parse_ply_points)."""
    import mmap
    with open(filename, "rb") as f:
        if _ply_is_binary(f, filename):
            return _load_binary_ply(filename)[0]             # an extension: the reference reads ASCII only
        size = os.fstat(f.fileno()).st_size
        # the parser's threads read the page cache through a mapping (f.read() copies the 10 MB of a vox10 cloud first: 1 ms)
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
    try:
        buf = np.frombuffer(mm, np.uint8) if size else np.zeros(0, np.uint8)
        cap = buf.size // 6 + 1                              # a point line is at least "0 0 0\n" (counting newlines costs 3 ms per 12 MB)
        out = np.empty((cap, 3), np.int32)                   # (pages the points never reach are never touched)
        n = np.zeros(1, np.int64)
        _lib.check_host(_lib.host().pcgc_parse_ply_points(_lib.nptr(buf) if buf.size else None, buf.size, _lib.nptr(out), cap,
                                                          _lib.nptr(n), min(64, _lib.host_threads())), "pcgc_parse_ply_points")
        del buf
    finally:
        if mm is not None:
            try:
                mm.close()
            except BufferError:                              # an error above left the array view alive: the GC unmaps
                pass
    return out[:int(n[0])]


def _ply_is_binary(f, filename):
    """Reads the header of an open ply file up to end_header, however long it is (comment / obj_info lines), and says
    whether it declares a binary format.  A file that starts with the `ply` magic but has no end_header is refused (the
    ASCII parser would return whatever lines happen to parse).  Files without the magic go to the ASCII parser like the
    reference's loop (inout_points.py:8-28: every line that parses is a point)."""
    head = f.read(4096)
    if not head.startswith(b"ply"):
        return False
    while b"end_header" not in head:
        more = f.read(65536)
        if not more:
            raise ValueError("%s: ply header without end_header" % filename)
        head += more
    return b"format binary_" in head[:head.find(b"end_header")]


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _load_binary_ply(filename):
    """binary_little_endian / binary_big_endian ply (what MeshLab, Open3D and CloudCompare write by default) ->
    (points int32 [N,3] truncated like the ASCII path, normals float32 [N,3] or None).  Vertex properties must be scalars
    (x, y, z and anything else: colours, normals); elements after the vertices are ignored, elements before them refused."""
    with open(filename, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    nl = data.find(b"\n", end)
    if end < 0 or nl < 0:
        raise ValueError("%s: ply header without end_header" % filename)
    order, element, n_vertex, fields, before = "<", None, None, [], 0
    for ln in data[:end].decode("ascii", "replace").splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "format":
            order = ">" if t[1] == "binary_big_endian" else "<"
        elif t[0] == "element":
            element = t[1]
            if element == "vertex":
                n_vertex = int(t[2])
            elif n_vertex is None and int(t[2]) > 0:
                before += 1
        elif t[0] == "property" and element == "vertex":
            if t[1] == "list" or t[1] not in _PLY_TYPES:
                raise ValueError("%s: vertex property %r is not a scalar of a known type" % (filename, " ".join(t[1:])))
            fields.append((t[2], order + _PLY_TYPES[t[1]]))
    if n_vertex is None or before or not all(k in dict(fields) for k in "xyz"):
        raise ValueError("%s: binary ply needs a leading vertex element with x, y, z" % filename)
    v = np.frombuffer(data, dtype=np.dtype(fields), count=n_vertex, offset=nl + 1)
    pts = np.stack([v["x"], v["y"], v["z"]], -1).astype(np.float64).astype(np.int32)
    nrm = None
    if all(k in v.dtype.names for k in ("nx", "ny", "nz")):
        nrm = np.stack([v["nx"], v["ny"], v["nz"]], -1).astype(np.float32)
    return pts, nrm


def load_ply_normals(filename):
    """ASCII ply with optional per-vertex normals -> (points int32 [N,3], normals float32 [N,3] or None).
    The reference hands the input ply to pc_error as its own normals file (eval.py:163, pc_error_wrapper.py:46-51);
    the property order is taken from the header."""
    with open(filename, "rb") as f:
        data = f.read()
    head_end = data.find(b"end_header")
    if head_end < 0:
        return load_ply_data(filename), None
    if b"format binary_" in data[:head_end]:
        return _load_binary_ply(filename)
    props = [ln.split()[-1].decode() for ln in data[:head_end].split(b"\n") if ln.strip().startswith(b"property")]
    nl = data.find(b"\n", head_end)
    body = data[nl + 1:] if nl >= 0 else b""
    if not body.strip():
        return np.zeros((0, 3), np.int32), None
    arr = np.loadtxt(io.BytesIO(body), dtype=np.float64, ndmin=2)
    col = {name: i for i, name in enumerate(props)}
    pts = arr[:, [col["x"], col["y"], col["z"]]].astype(np.int32)
    if all(k in col for k in ("nx", "ny", "nz")):
        return pts, arr[:, [col["nx"], col["ny"], col["nz"]]].astype(np.float32)
    return pts, None


class _Text(__import__("threading").local):
    """reused formatting buffer (a fresh 12 MB array costs its page faults per cloud) — one per THREAD: the body returned by
    _ply_parts is a view of it that the caller is still writing to disk when another thread formats its own cloud"""

    def __init__(self):
        self.buf = np.empty(0, np.uint8)


_TEXT = _Text()


def ply_header(n):
    return ("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "end_header\n" % n).encode()


def _ply_parts(points):
    """-> (header bytes, body as a bytes-like object): exactly the text write_ply_data (inout_points.py:30-46) produces."""
    points = np.asarray(points)
    head = ply_header(points.shape[0])
    if points.shape[0] == 0:
        return head, b""
    if np.issubdtype(points.dtype, np.integer):
        pts = np.ascontiguousarray(points[:, :3], np.int64)
        n = np.zeros(1, np.int64)
        host = _lib.host()
        buf = _TEXT.buf
        rc = host.pcgc_format_points_int(_lib.nptr(pts), pts.shape[0], _lib.nptr(buf) if buf.size else None, buf.size, _lib.nptr(n))
        if rc == -2:                                          # too small: *out_len holds the size this cloud needs
            buf = _TEXT.buf = np.empty(int(n[0]), np.uint8)
            rc = host.pcgc_format_points_int(_lib.nptr(pts), pts.shape[0], _lib.nptr(buf), buf.size, _lib.nptr(n))
        _lib.check_host(rc, "pcgc_format_points_int")
        return head, memoryview(buf)[:int(n[0])]
    body = ply_bytes(points)
    return b"", body


def ply_bytes(points):
    """Exactly the text write_ply_data (inout_points.py:30-46) produces: header + str() of each coordinate."""
    points = np.asarray(points)
    head = ("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "end_header\n" % points.shape[0])
    if points.shape[0] == 0:
        return head.encode()
    if np.issubdtype(points.dtype, np.integer):
        h, b = _ply_parts(points)
        return h + bytes(b)
    else:
        s = np.array([[str(v) for v in row] for row in points]) if points.shape[0] < 64 else _float_str(points)
    lines = np.char.add(np.char.add(np.char.add(np.char.add(s[:, 0], " "), s[:, 1]), " "), s[:, 2])
    return (head + "\n".join(lines.tolist()) + "\n").encode()


def _float_str(points):
    flat = points.reshape(-1)
    uniq, inv = np.unique(flat, return_inverse=True)
    table = np.array([str(v) for v in uniq])            # numpy scalar str(): shortest round-trip repr, e.g. '12.0'
    return table[inv].reshape(points.shape)


def write_ply_data(filename, points):
    head, body = _ply_parts(points)
    with open(filename, "wb") as f:
        f.write(head)
        f.write(body)


# ---------------------------------------------------------------------------- partition
def _order_key(cube_positions):
    cube_positions = np.asarray(cube_positions).astype(np.int64)
    step = cube_positions.max() + 1
    return cube_positions[:, 0] + cube_positions[:, 1] * step + cube_positions[:, 2] * step * step, step


def ordered_positions(cube_positions):
    """Cube positions in the order the cubes are stored (inout_points.py:80-86, 96-102)."""
    key, step = _order_key(cube_positions)
    key = np.sort(key)
    return np.stack([key % step, (key // step) % step, key // step // step], -1)


def partition(points, cube_size=64, min_num=20):
    """Core of load_points on an in-memory cloud.  Returns
    (cube_positions [first-appearance order], sorted_positions, cube_of_point [index into sorted cubes, -1 dropped])."""
    points = np.ascontiguousarray(points, np.int32).reshape(-1, 3)
    n = points.shape[0]
    lib = _lib.host()
    ncub = np.zeros(1, np.int64)
    cap = n // max(int(min_num), 1) + 1                  # a kept cube holds >= min_num points: one call, buffers at the bound
    pos = np.empty((cap, 3), np.int64)
    spos = np.empty((cap, 3), np.int64)
    cop = np.empty(n, np.int32)
    rc = lib.pcgc_partition(_lib.nptr(points), n, cube_size, min_num, _lib.nptr(ncub), _lib.nptr(pos), _lib.nptr(spos), _lib.nptr(cop))
    if rc == -4 or (rc == 0 and int(ncub[0]) == 0):
        raise ValueError("no cube holds at least min_num=%d points" % min_num)
    _lib.check_host(rc, "pcgc_partition")
    B = int(ncub[0])
    pos, spos = pos[:B].copy(), spos[:B].copy()
    return pos, spos, cop


def load_points(filename, cube_size=64, min_num=20):
    """-> (set_points: list of int16 [n_i,3] in stored-cube order, cube_positions in first-appearance order)."""
    pts = load_ply_data(filename)
    pos, spos, cop = partition(pts, cube_size, min_num)
    keep = cop >= 0
    order = np.argsort(cop[keep], kind="stable")
    local = (pts[keep] % cube_size).astype(np.int16)[order]
    counts = np.bincount(cop[keep], minlength=len(pos))
    return np.split(local, np.cumsum(counts)[:-1]), pos


def save_points(set_points, cube_positions, filename, cube_size=64):
    write_ply_data(filename, merge_points(set_points, cube_positions, cube_size))


def merge_points(set_points, cube_positions, cube_size=64):
    spos = ordered_positions(cube_positions)
    out = [np.asarray(v) + spos[i] * cube_size for i, v in enumerate(set_points)]
    return np.concatenate(out).astype("int")


def voxels2merged_points(voxels, cube_positions, cube_size=64, ordered=False):
    """merge_points(voxels2points(voxels), cube_positions, cube_size) for a device tensor in one pass: the global
    coordinates are formed on the GPU (voxel index + sorted cube position * cube_size) and come to the host as one
    C-contiguous int64 [n, 3] array in the same order (cubes in list order, row-major inside a cube).  ordered=True:
    cube_positions[i] already is the position of voxels[i] (a slice of ordered_positions, process.StreamedPostprocess)."""
    import torch
    v = voxels.reshape(voxels.shape[:4])
    idx = torch.nonzero(v > 0)                           # [n, 4] int64: cube, x, y, z — sorted lexicographically
    spos = cube_positions if torch.is_tensor(cube_positions) else torch.from_numpy(np.ascontiguousarray(
        cube_positions if ordered else ordered_positions(cube_positions), np.int64))
    spos = spos.to(idx.device)
    # written into a row-major buffer: the sum of a column slice and a gather comes out column-major, and turning
    # 20 MB around on the host afterwards cost 5 ms per cloud
    out = torch.empty((idx.shape[0], 3), dtype=torch.int64, device=idx.device)
    torch.add(idx[:, 1:], spos[idx[:, 0]] * int(cube_size), out=out)
    # through page-locked memory: a copy into pageable memory is staged in small pieces (0.4 ms for the 3 MB of a decoder slice)
    host = torch.empty((idx.shape[0], 3), dtype=torch.int64, pin_memory=True)
    host.copy_(out, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


# ---------------------------------------------------------------------------- voxels
def points2voxels(set_points, cube_size, device=True):
    """list of [n_i,3] local points -> occupancy cubes [B,cs,cs,cs,1].  device=True: float32 torch tensor in HBM
    (scatter by `pcgc_voxelize`); device=False: float64 numpy like the reference."""
    B = len(set_points)
    lens = np.array([len(p) for p in set_points], np.int64)
    cube_idx = np.repeat(np.arange(B, dtype=np.int32), lens)
    xyz = np.concatenate([np.asarray(p, np.int32).reshape(-1, 3) for p in set_points]) if B else np.zeros((0, 3), np.int32)
    return voxelize(cube_idx, xyz, B, cube_size, device)


def voxelize(cube_idx, xyz, B, cube_size, device=True):
    import torch
    dev = _lib.require_gpu()
    rec = np.empty((len(cube_idx), 4), np.int32)
    rec[:, 0] = cube_idx
    rec[:, 1:] = xyz
    rec_d = torch.from_numpy(rec).to(dev)
    cubes = torch.zeros((B, cube_size, cube_size, cube_size, 1), dtype=torch.float32, device=dev)
    _lib.check(_lib.hip().pcgc_voxelize(_lib.dptr(rec_d), rec.shape[0], cube_size, _lib.dptr(cubes), B, _lib.stream()),
               "pcgc_voxelize")
    if device:
        return cubes
    return cubes.cpu().numpy().astype(np.float64)


def voxelize_partition(points, cube_of_point, lo, hi, cube_size):
    """Device cubes [hi - lo, cs, cs, cs, 1] of the key-sorted cubes lo <= c < hi straight from partition()'s outputs:
    the points and their cube indices go up as they are (13 B per point), the kernel drops the points of other cubes
    and takes the coordinates mod cube_size — no per-point records are built on the host."""
    import torch
    dev = _lib.require_gpu()
    pts_d = torch.from_numpy(np.ascontiguousarray(points, np.int32)).to(dev, non_blocking=True)
    cop_d = torch.from_numpy(np.ascontiguousarray(cube_of_point, np.int32)).to(dev, non_blocking=True)
    cubes = torch.zeros((hi - lo, cube_size, cube_size, cube_size, 1), dtype=torch.float32, device=dev)
    _lib.check(_lib.hip().pcgc_voxelize_points(_lib.dptr(pts_d), _lib.dptr(cop_d), int(pts_d.shape[0]), cube_size, int(lo), int(hi),
                                               _lib.dptr(cubes), _lib.stream()), "pcgc_voxelize_points")
    return cubes


def voxels2points(voxels):
    """0/1 volumes [B,cs,cs,cs(,1)] -> list of int [n_i,3] in row-major order (np.where order)."""
    import torch
    if torch.is_tensor(voxels):
        v = voxels.reshape(voxels.shape[:4])
        idx = torch.nonzero(v > 0)                       # sorted lexicographically = row-major
        counts = torch.bincount(idx[:, 0], minlength=v.shape[0]).cpu().numpy()
        pts = idx[:, 1:].cpu().numpy()
        return np.split(pts, np.cumsum(counts)[:-1])
    v = np.uint8(np.asarray(voxels))
    v = v.reshape(v.shape[:4])
    return [np.array(np.where(c > 0)).transpose((1, 0)) for c in v]


def select_voxels(vols, points_nums, offset_ratio=1.0, fixed_thres=None, return_thresholds=False):
    """Adaptive top-k mask on the device.  vols: torch cuda / numpy [B,cs,cs,cs,1] float32 -> mask uint8 torch
    tensor of the same shape (1 = selected)."""
    import torch
    dev = _lib.require_gpu()
    x = vols if torch.is_tensor(vols) else torch.from_numpy(np.ascontiguousarray(vols, np.float32))
    x = x.to(dev, torch.float32).contiguous()
    B = int(x.shape[0])
    vox = x.numel() // max(B, 1)
    k = np.array([int(offset_ratio * np.array(n)) for n in np.asarray(points_nums).reshape(-1)], np.int32)
    assert len(k) == B
    k_d = torch.from_numpy(k).to(dev)
    thr = torch.empty(B, dtype=torch.float32, device=dev)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    _lib.check(_lib.hip().pcgc_topk_threshold(_lib.dptr(x), _lib.dptr(k_d), B, vox, int(fixed_thres is not None),
                                              float(fixed_thres if fixed_thres is not None else 0.0), _lib.dptr(thr),
                                              _lib.dptr(mask), None, 0, _lib.stream()), "pcgc_topk_threshold")
    return (mask, thr) if return_thresholds else mask


def get_adaptive_thres(vol, num, init_thres=-2.0):
    assert init_thres == -2.0, "the device kernel implements the reference's init_thres=-2.0"
    import torch
    v = vol if torch.is_tensor(vol) else torch.from_numpy(np.ascontiguousarray(vol, np.float32))
    _, thr = select_voxels(v.reshape((1,) + tuple(v.shape)), [num], 1.0, return_thresholds=True)
    return float(thr[0])
