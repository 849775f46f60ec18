"""Minimal pure-Python reader for the HDF5 files of the reference's training set (generate_dataset.py:11-38: one file
per cube, `h5py.File(name, 'w').create_dataset('data', data=points, shape=points.shape)` with points uint8 [n, 3]) —
used by train_hyper.load_cube_points when h5py is not installed (it is not part of this image).

Covers what h5py's defaults produce for such a file (HDF5 File Format Specification 3.0, "earliest" library format):
  superblock version 0 / 1            -> root group symbol-table entry
  version-1 object headers            (+ continuation blocks)
  old-style groups                    symbol table message -> v1 B-tree (TREE) -> symbol nodes (SNOD) + local heap (HEAP)
  dataset messages                    dataspace v1 / v2 (simple), datatype class 0 / 1 (fixed / floating point, little
                                      endian, 1-8 bytes), data layout v3 contiguous or compact
Anything else (chunked / filtered datasets, version-2 object headers of libver='latest' files, big-endian types) raises
NotImplementedError naming the feature — never a wrong array.  *** PARITY UNPINNED *** against h5py-written files (none
exist offline); pinned to tests/golden/cube_points.h5, assembled byte by byte from the specification by
tools/make_h5_fixture.py, which does not import this module.
"""
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File(object):
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != _SIG:
            raise ValueError("not an HDF5 file (signature)")
        ver = buf[8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (libver='latest' files) is not supported by the minimal reader" % ver)
        if buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError("HDF5 offsets / lengths of %d / %d bytes" % (buf[13], buf[14]))
        pos = 24 + (4 if ver == 1 else 0)                       # v1 adds indexed-storage K + reserved
        self.base, _free, _eof, _drv = struct.unpack_from("<4Q", buf, pos)
        pos += 32
        _name_off, self.root_header, cache, _r = struct.unpack_from("<QQII", buf, pos)
        self.root_scratch = struct.unpack_from("<QQ", buf, pos + 24) if cache == 1 else None

    # ---------------------------------------------------------------- object headers (version 1)
    def messages(self, addr):
        """-> [(type, payload bytes)] of the version-1 object header at addr, following continuation messages"""
        b = self.b
        addr += self.base
        if b[addr:addr + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (libver='latest') are not supported by the minimal reader")
        if b[addr] != 1:
            raise ValueError("object header version %d at %d" % (b[addr], addr))
        nmsg, _refs, size = struct.unpack_from("<HII", b, addr + 2)
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                out.append((mtype, body))
                if mtype == 0x10:                                # continuation: offset, length
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    blocks.append((self.base + off, ln))
        return out

    # ---------------------------------------------------------------- old-style groups
    def group_links(self, btree, heap):
        """name -> object header address for a symbol-table group"""
        b = self.b
        heap += self.base
        if b[heap:heap + 4] != b"HEAP":
            raise ValueError("local heap signature")
        heap_data = self.base + struct.unpack_from("<Q", b, heap + 24)[0]
        links = {}

        def node(addr):
            addr += self.base
            if b[addr:addr + 4] != b"TREE":
                raise ValueError("B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", b, addr + 4)
            if ntype != 0:
                raise ValueError("group B-tree expected")
            pos = addr + 24                                      # past left / right sibling
            for i in range(used):
                child = struct.unpack_from("<Q", b, pos + 8)[0]  # key i (8), child i (8)
                pos += 16
                if level > 0:
                    node(child)
                else:
                    snod(child)

        def snod(addr):
            addr += self.base
            if b[addr:addr + 4] != b"SNOD":
                raise ValueError("symbol node signature")
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                name_off, hdr = struct.unpack_from("<QQ", b, addr + 8 + 40 * i)
                p = heap_data + name_off
                links[bytes(b[p:b.index(b"\0", p)]).decode()] = hdr
        node(btree)
        return links

    def root_links(self):
        if self.root_scratch is not None:
            return self.group_links(*self.root_scratch)
        for t, body in self.messages(self.root_header):
            if t == 0x11:
                return self.group_links(*struct.unpack_from("<QQ", body, 0))
        raise NotImplementedError("root group without a symbol table (new-style groups)")

    # ---------------------------------------------------------------- datasets
    def dataset(self, addr):
        shape = dtype = layout = None
        for t, body in self.messages(addr):
            if t == 0x01:                                        # dataspace
                ver, rank, flags = body[0], body[1], body[2]
                off = 8 if ver == 1 else 4
                if ver not in (1, 2) or (ver == 2 and body[3] not in (0, 1)):
                    raise NotImplementedError("dataspace message version %d / type %d" % (ver, body[3]))
                shape = struct.unpack_from("<%dQ" % rank, body, off) if rank else ()
            elif t == 0x03:                                      # datatype
                cls, bits0, size = body[0] & 0x0F, body[1], struct.unpack_from("<I", body, 4)[0]
                if bits0 & 1:
                    raise NotImplementedError("big-endian datatype")
                if cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 0x08 else "u", size))
                elif cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype("<f%d" % size)
                else:
                    raise NotImplementedError("datatype class %d of %d bytes" % (cls, size))
            elif t == 0x08:                                      # data layout
                if body[0] != 3:
                    raise NotImplementedError("data layout message version %d" % body[0])
                if body[1] == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", body, 2)
                elif body[1] == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    layout = ("compact", bytes(body[4:4 + n]))
                else:
                    raise NotImplementedError("chunked dataset layout (written with chunks= / compression=)")
            elif t == 0x0B:
                raise NotImplementedError("filtered (compressed) dataset")
        if shape is None or dtype is None or layout is None:
            raise ValueError("object at %d is not a dataset" % addr)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if layout[0] == "compact":
            raw = layout[1]
        elif layout[1] == _UNDEF:                                # never written: fill value zero
            return np.zeros(shape, dtype)
        else:
            raw = self.b[self.base + layout[1]:self.base + layout[1] + layout[2]]
        if len(raw) < count * dtype.itemsize:
            raise ValueError("dataset holds %d bytes, %d expected" % (len(raw), count * dtype.itemsize))
        return np.frombuffer(raw, dtype, count).reshape(shape).copy()


def read_dataset(path, name="data"):
    """The array of dataset `name` in the root group of the HDF5 file at `path`."""
    with open(path, "rb") as f:
        h = _File(f.read())
    links = h.root_links()
    if name not in links:
        raise KeyError("%s: no dataset %r in the root group (has: %s)" % (path, name, sorted(links)))
    return h.dataset(links[name])


def write_dataset(path, array, name="data"):
    """Write `array` (uint8 / int / float, any rank) as the contiguous dataset `name` in the root group of a new HDF5 file:
    superblock 0, old-style root group (B-tree + symbol node + local heap), version-1 dataset header with dataspace,
    datatype, fill-value and data-layout messages — the structures h5py's default ("earliest") format uses for
    generate_dataset.py:27-29.  Read back by read_dataset here; meant to be readable by libhdf5 / h5py (unverified offline)."""
    a = np.ascontiguousarray(array)
    if a.dtype.kind not in "uif" or a.dtype.itemsize not in (1, 2, 4, 8):
        raise ValueError("dtype %s cannot be written" % a.dtype)
    a = a.astype(a.dtype.newbyteorder("<"))
    raw = a.tobytes()
    nm = name.encode() + b"\0"
    nm += b"\0" * (-len(nm) % 8)

    def msg(mtype, body):
        body = body + b"\0" * (-len(body) % 8)
        return struct.pack("<HHB3x", mtype, len(body), 0) + body
    heap_size = max(88, 8 + len(nm) + 16)
    heap_size += -heap_size % 8
    a_root, a_tree = 96, 136
    a_snod = a_tree + 24 + 33 * 8 + 32 * 8
    a_heap = a_snod + 8 + 8 * 40
    a_heapdata = a_heap + 32
    a_dset = a_heapdata + heap_size
    dspace = bytes([1, a.ndim, 0, 0, 0, 0, 0, 0]) + struct.pack("<%dQ" % a.ndim, *a.shape)
    if a.dtype.kind == "f":
        mant = {2: 10, 4: 23, 8: 52}[a.dtype.itemsize]
        expo = {2: 5, 4: 8, 8: 11}[a.dtype.itemsize]
        dtype = bytes([0x11, 0x20, a.dtype.itemsize * 8 - 1, 0x00]) + struct.pack("<I", a.dtype.itemsize)
        dtype += struct.pack("<HHBBBBI", 0, a.dtype.itemsize * 8, mant, expo, 0, mant, (1 << (expo - 1)) - 1)
    else:
        dtype = bytes([0x10, 0x08 if a.dtype.kind == "i" else 0x00, 0, 0]) + struct.pack("<I", a.dtype.itemsize)
        dtype += struct.pack("<HH", 0, a.dtype.itemsize * 8)
    d_msgs = msg(0x0001, dspace) + msg(0x0003, dtype) + msg(0x0005, bytes([2, 2, 2, 0]))
    a_raw = a_dset + 16 + len(d_msgs) + 8 + 24
    d_msgs += msg(0x0008, bytes([3, 1]) + struct.pack("<QQ", a_raw if raw else _UNDEF, len(raw)))
    dset = struct.pack("<BxHII4x", 1, 4, 1, len(d_msgs)) + d_msgs
    assert a_dset + len(dset) == a_raw
    eof = a_raw + len(raw)
    sb = _SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", 4, 16, 0) + struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_tree, a_heap)
    root_msgs = msg(0x0011, struct.pack("<QQ", a_tree, a_heap))
    root = struct.pack("<BxHII4x", 1, 1, 1, len(root_msgs)) + root_msgs
    tree = b"TREE" + struct.pack("<BBH", 0, 0, 1) + struct.pack("<QQ", _UNDEF, _UNDEF) + struct.pack("<QQQ", 0, a_snod, 8)
    tree += b"\0" * (a_snod - a_tree - len(tree))
    snod = b"SNOD" + struct.pack("<BxH", 1, 1) + struct.pack("<QQII16x", 8, a_dset, 0, 0)
    snod += b"\0" * (a_heap - a_snod - len(snod))
    free_off = 8 + len(nm)
    heap_data = b"\0" * 8 + nm + struct.pack("<QQ", 1, heap_size - free_off)
    heap_data += b"\0" * (heap_size - len(heap_data))
    heap = b"HEAP" + bytes(4) + struct.pack("<QQQ", heap_size, free_off, a_heapdata) + heap_data
    with open(path, "wb") as f:
        f.write(sb + root + tree + snod + heap + dset + raw)
