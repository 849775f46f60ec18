"""Hand-assembles tests/golden/cube_points.h5 from the HDF5 File Format Specification (version 3.0, the "earliest" library
format h5py writes by default): the layout of one training-set file of the reference (generate_dataset.py:27-29: dataset
'data', uint8 [n, 3]).  Does NOT import pcgcv1_amd/dataprocess/h5min.py — the reader is tested against these bytes.

    python tools/make_h5_fixture.py

File map (all addresses relative to base 0):
     0  superblock version 0 (56 bytes) + root group symbol-table entry (40 bytes, cache type 1: B-tree + heap addresses)
    96  root group object header, version 1: one symbol-table message (type 0x0011)
   136  v1 B-tree node "TREE" (group node, level 0, 1 entry -> the symbol node); full node size for K = 16
   680  symbol node "SNOD" version 1 with 1 symbol ('data'); room for 2 * leaf K = 8 entries
  1008  local heap "HEAP": header 32 bytes, data segment of 88 bytes at 1040: "" at 0, "data" at 8
  1128  dataset object header, version 1 (16 + 96 bytes): dataspace v1 rank 2, datatype fixed-point unsigned 1 byte,
        fill value v2, continuation -> block at 1240 holding the data layout v3 message (contiguous) [5 messages with it]
  1240  continuation block (32 bytes)
  1272  raw data: n * 3 bytes
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "cube_points.h5")
UNDEF = 0xFFFFFFFFFFFFFFFF


def points():
    """the cube's points: a seeded walk inside a 64^3 cube, uint8 [37, 3]"""
    out, s = [], 12345
    for _ in range(37):
        row = []
        for _c in range(3):
            s = (s * 1103515245 + 12345) & 0x7FFFFFFF
            row.append((s >> 16) % 64)
        out.append(row)
    return out


def msg(mtype, body):
    body = body + b"\0" * (-len(body) % 8)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def main():
    pts = points()
    raw = bytes(v for row in pts for v in row)
    A_ROOT, A_TREE, A_SNOD, A_HEAP = 96, 136, 680, 1008
    A_HEAPDATA, A_DSET, A_CONT, A_RAW = 1040, 1128, 1240, 1272
    eof = A_RAW + len(raw)
    sb = b"\x89HDF\r\n\x1a\n" + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, A_ROOT, 1, 0) + struct.pack("<QQ", A_TREE, A_HEAP)
    assert len(sb) == 96
    root_msgs = msg(0x0011, struct.pack("<QQ", A_TREE, A_HEAP))
    root = struct.pack("<BxHII4x", 1, 1, 1, len(root_msgs)) + root_msgs
    assert len(root) == 40
    tree = b"TREE" + struct.pack("<BBH", 0, 0, 1) + struct.pack("<QQ", UNDEF, UNDEF)
    tree += struct.pack("<QQQ", 0, A_SNOD, 8)                       # key 0 (heap offset of ""), child 0, key 1 ("data")
    tree += b"\0" * (24 + 33 * 8 + 32 * 8 - len(tree))
    assert A_TREE + len(tree) == A_SNOD
    snod = b"SNOD" + struct.pack("<BxH", 1, 1) + struct.pack("<QQII16x", 8, A_DSET, 0, 0)
    snod += b"\0" * (8 + 8 * 40 - len(snod))
    assert A_SNOD + len(snod) == A_HEAP
    heap_data = b"\0" * 8 + b"data\0\0\0\0" + struct.pack("<QQ", 1, 72)       # names, then one free block (next = 1 = none, size)
    heap_data += b"\0" * (88 - len(heap_data))
    heap = b"HEAP" + bytes([0, 0, 0, 0]) + struct.pack("<QQQ", 88, 16, A_HEAPDATA) + heap_data
    assert A_HEAP + len(heap) == A_DSET
    dspace = bytes([1, 2, 0, 0, 0, 0, 0, 0]) + struct.pack("<QQ", len(pts), 3)
    dtype = bytes([0x10, 0x00, 0x00, 0x00]) + struct.pack("<I", 1) + struct.pack("<HH", 0, 8)   # v1, class 0, LE, unsigned
    fill = bytes([2, 2, 2, 0])                                       # version 2, late allocation, fill if set, undefined
    cont = struct.pack("<QQ", A_CONT, 32)
    d_msgs = msg(0x0001, dspace) + msg(0x0003, dtype) + msg(0x0005, fill) + msg(0x0010, cont)
    dset = struct.pack("<BxHII4x", 1, 5, 1, len(d_msgs)) + d_msgs
    assert A_DSET + len(dset) == A_CONT, (A_DSET + len(dset), A_CONT)
    layout = bytes([3, 1]) + struct.pack("<QQ", A_RAW, len(raw))
    block = msg(0x0008, layout)
    block += msg(0x0000, b"\0" * (32 - len(block) - 8)) if len(block) < 32 else b""
    assert len(block) == 32 and A_CONT + len(block) == A_RAW
    data = sb + root + tree + snod + heap + dset + block + raw
    assert len(data) == eof
    with open(OUT, "wb") as f:
        f.write(data)
    print("wrote %s (%d bytes, %d points)" % (OUT, len(data), len(pts)))
    return pts


if __name__ == "__main__":
    main()
