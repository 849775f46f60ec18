"""Hand-assembles a minimal TensorFlow tensor bundle, byte by byte, from the published on-disk formats — WITHOUT using
pcgcv1_amd/tf_bundle.py — and writes it to tests/golden/tf_bundle_min.{index,data-00000-of-00001}.

    python tools/make_tf_bundle_fixture.py

The reader (tf_bundle.read_bundle / checkpoint loading) must parse these files and the writer (tf_bundle.write_bundle)
must reproduce them byte for byte (tests/test_host_cpu.py::test_hand_assembled_bundle).  No TensorFlow exists offline,
so this pins the two implementations (this script, tf_bundle.py) against each other and against the format documents,
not against TensorFlow's own output.

Formats followed (TensorFlow 1.13):
  tensorflow/core/util/tensor_bundle/tensor_bundle.cc   BundleWriter: data file = tensor bytes back to back in Add() order,
                                                        index = io::Table (kNoCompression), key "" -> BundleHeaderProto
  tensorflow/core/protobuf/tensor_bundle.proto          BundleHeaderProto{1 num_shards, 2 endianness, 3 version{1 producer}}
                                                        BundleEntryProto{1 dtype, 2 shape, 3 shard_id, 4 offset, 5 size,
                                                                         6 fixed32 crc32c (masked)}
  tensorflow/core/framework/tensor_shape.proto          TensorShapeProto{2 repeated Dim{1 int64 size}}
  tensorflow/core/framework/types.proto                 DT_FLOAT = 1, DT_STRING = 7, DT_INT64 = 9
  tensorflow/core/lib/io/{table_builder,block_builder,format}.cc  (= leveldb's table format)
      block   : entries [varint shared][varint non_shared][varint value_len][key suffix][value], restart every 16 entries,
                then uint32 restart offsets, uint32 restart count
      trailer : 1 byte compression type (0) + uint32 masked crc32c(block + type byte)
      index   : one entry per data block, key = short successor of the block's last key, value = BlockHandle(varint
                offset, varint size); restart interval 1
      footer  : metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (little endian)
  tensorflow/core/lib/hash/crc32c.h                     mask(crc) = rotr(crc, 15) + 0xa282ead8
  tensorflow/core/protobuf/checkpointable_object_graph.proto
      CheckpointableObjectGraph{1 repeated nodes{1 repeated children{1 node_id, 2 local_name},
                                                 2 repeated attributes{1 name, 2 full_name, 3 checkpoint_key}}}
  DT_STRING tensors on disk (tensor_bundle.cc WriteStringTensor): [varint length per element][uint32 masked crc32c of the
      lengths as uint64s][element bytes]; the entry's crc covers lengths (as uint64), the length checksum and the bytes.
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "tf_bundle_min")


def crc32c(data, crc=0):                      # Castagnoli, reflected polynomial 0x82F63B78, bit by bit
    crc ^= 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def tag(field, wire):
    return varint((field << 3) | wire)


def ldelim(field, payload):
    return tag(field, 2) + varint(len(payload)) + payload


def main():
    # ---------------------------------------------------------------- the two tensors
    # estimator/matrix_0: float32 [2, 3, 1] = 0.5, -1.25, 2, 0, 3.75, -8      global_step: int64 scalar 5000
    matrix = struct.pack("<6f", 0.5, -1.25, 2.0, 0.0, 3.75, -8.0)
    step = struct.pack("<q", 5000)
    k_matrix = b"estimator/matrix_0/.ATTRIBUTES/VARIABLE_VALUE"
    k_step = b"global_step/.ATTRIBUTES/VARIABLE_VALUE"
    k_graph = b"_CHECKPOINTABLE_OBJECT_GRAPH"

    # ---------------------------------------------------------------- object graph (breadth first: 0 root, 1 estimator,
    # 2 global_step, 3 matrix_0)
    def node(children, attribute=None):
        body = b""
        for name, nid in children:
            body += ldelim(1, tag(1, 0) + varint(nid) + ldelim(2, name))
        if attribute:
            full, key = attribute
            body += ldelim(2, ldelim(1, b"VARIABLE_VALUE") + ldelim(2, full) + ldelim(3, key))
        return ldelim(1, body)
    graph = (node([(b"estimator", 1), (b"global_step", 2)]) + node([(b"matrix_0", 3)])
             + node([], (b"global_step", k_step)) + node([], (b"estimator/matrix_0", k_matrix)))

    # ---------------------------------------------------------------- data file: tensors in key order, then the graph string
    lengths = varint(len(graph))
    len_ck = struct.pack("<I", mask(crc32c(struct.pack("<Q", len(graph)))))
    graph_raw = lengths + len_ck + graph
    graph_crc = mask(crc32c(graph, crc32c(len_ck, crc32c(struct.pack("<Q", len(graph))))))
    data = matrix + step + graph_raw
    off_matrix, off_step, off_graph = 0, len(matrix), len(matrix) + len(step)

    # ---------------------------------------------------------------- index entries (values of the table)
    def dims(*sizes):
        return b"".join(ldelim(2, tag(1, 0) + varint(s)) for s in sizes)

    def entry(dtype, shape, offset, size, crc):
        out = tag(1, 0) + varint(dtype) + ldelim(2, shape)
        if offset:
            out += tag(4, 0) + varint(offset)
        return out + tag(5, 0) + varint(size) + tag(6, 5) + struct.pack("<I", crc)
    header = tag(1, 0) + varint(1) + ldelim(3, tag(1, 0) + varint(1))          # num_shards 1, version.producer 1
    table = sorted([
        (b"", header),
        (k_graph, entry(7, dims(), off_graph, len(graph_raw), graph_crc)),
        (k_matrix, entry(1, dims(2, 3, 1), off_matrix, len(matrix), mask(crc32c(matrix)))),
        (k_step, entry(9, dims(), off_step, len(step), mask(crc32c(step)))),
    ])

    # ---------------------------------------------------------------- one data block (4 entries < restart interval 16)
    block, last = b"", b""
    for key, value in table:
        shared = 0
        while shared < min(len(key), len(last)) and key[shared] == last[shared]:
            shared += 1
        block += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        last = key
    block += struct.pack("<I", 0) + struct.pack("<I", 1)                       # restart array [0], count 1

    def with_trailer(b):
        return b + b"\x00" + struct.pack("<I", mask(crc32c(b + b"\x00")))
    index_file = with_trailer(block)
    data_handle = varint(0) + varint(len(block))
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                         # empty block
    meta_handle = varint(len(index_file)) + varint(len(meta))
    index_file += with_trailer(meta)
    # index block: key = short successor of the last data key "global_step/..." = "h"
    succ = bytes([last[0] + 1])
    iblock = varint(0) + varint(len(succ)) + varint(len(data_handle)) + succ + data_handle
    iblock += struct.pack("<I", 0) + struct.pack("<I", 1)
    index_handle = varint(len(index_file)) + varint(len(iblock))
    index_file += with_trailer(iblock)
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    index_file += footer

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".index", "wb") as f:
        f.write(index_file)
    with open(OUT + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    print("wrote %s.index (%d bytes), .data-00000-of-00001 (%d bytes)" % (OUT, len(index_file), len(data)))


if __name__ == "__main__":
    main()
