"""TensorFlow tensor-bundle checkpoints without TensorFlow.

The reference restores its five sub-models with
    tf.train.Checkpoint(analysis_transform=..., synthesis_transform=..., hyper_encoder=..., hyper_decoder=...,
                        estimator=...).restore(tf.train.latest_checkpoint(ckpt_dir))      (transform.py:107-112, 214-218)
and saves them with checkpoint.save(file_prefix=".../ckpt") (train_hyper.py:255-268).  On disk that is
  <dir>/checkpoint                      text proto: model_checkpoint_path: "ckpt-N"
  <dir>/ckpt-N.index                    LevelDB-format table: "" -> BundleHeaderProto, name -> BundleEntryProto
  <dir>/ckpt-N.data-00000-of-00001      raw little-endian tensor bytes at (offset, size)
Object-based checkpoints name a variable by its attribute path plus "/.ATTRIBUTES/VARIABLE_VALUE"
(e.g. "analysis_transform/vrn1_1/conv1_1/kernel/.ATTRIBUTES/VARIABLE_VALUE", "estimator/bais_0/...");
`read_bundle` strips that suffix, so name-based (tf.train.Saver) bundles load the same way.

Format restated from TensorFlow 1.13 (tensorflow/core/util/tensor_bundle, core/lib/io/{table,block,format},
core/protobuf/tensor_bundle.proto, core/framework/{tensor_shape,types}.proto) — the package is absent here and the
reference tree holds no checkpoint files, so this reader is *** PARITY UNPINNED *** against real TF output; what is
pinned: CRC-32C known answers, the LevelDB table magic, a write -> read round trip (tests/test_host_cpu.py), and a small
bundle assembled byte by byte from the published formats by an independent script (tools/make_tf_bundle_fixture.py ->
tests/golden/tf_bundle_min.*) that the reader parses and the writer reproduces byte for byte.

Object graph.  tf.train.Checkpoint.restore() does not look variables up by key: it walks the serialized
`CheckpointableObjectGraph` (tensorflow/core/protobuf/checkpointable_object_graph.proto, TF 1.13 naming) stored as the
DT_STRING scalar "_CHECKPOINTABLE_OBJECT_GRAPH", matching nodes to live objects edge by edge (children[].local_name =
the attribute name under which Keras tracks the sub-object: `conv_in`, `vrn1_1` -> `conv1_1`, ... -> `kernel` / `bias`;
EntropyBottleneck.add_variable tracks `matrix_0`, `bais_0`, `factor_0`, ... by variable name, entropy_model.py:51-66)
and reading each node's attributes[].checkpoint_key.  `object_graph()` builds that proto from the variable paths, so a
bundle written here carries what the reference's restore needs (transform.py:107-112, train_hyper.py:107-121, 275-284):
    node 0 = root; edge per path component; leaf nodes carry SerializedTensor{name "VARIABLE_VALUE", full_name = path,
    checkpoint_key = path + "/.ATTRIBUTES/VARIABLE_VALUE"}.
Assumptions that cannot be checked offline: Keras 1.13 also adds "layer-N" / "layer_with_weights-N" edges for a
Network's layers (extra edges in the LIVE object are harmless for restore(); the reference never calls
assert_consumed()), and Adam's slot variables are referenced from the optimizer node (not written: the reference's
default --reset_optimizer=0 path saves no optimizer, train_hyper.py:107-113).
"""
import os
import re
import struct

import numpy as np

from . import _lib

_MAGIC = 0xDB4775248B80FB57
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}
_DT_STRING = 7


def crc32c(data, crc=0):
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
    buf = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return int(_lib.host().pcgc_crc32c(crc, buf.ctypes.data if buf.size else None, buf.size))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / minimal protobuf
def _get_varint(b, pos):
    out = shift = 0
    while True:
        c = b[pos]
        pos += 1
        out |= (c & 0x7F) << shift
        if c < 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _pb_fields(b):
    """Yield (field number, wire type, value) of one protobuf message."""
    pos = 0
    while pos < len(b):
        key, pos = _get_varint(b, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(b, pos)
        elif wt == 1:
            v = b[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(b, pos)
            v = b[pos:pos + n]
            pos += n
        elif wt == 5:
            v = b[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(b):
    dims = []
    for f, _, v in _pb_fields(b):
        if f == 2:                                  # repeated Dim
            size = 0
            for g, _, u in _pb_fields(v):
                if g == 1:
                    size = _signed(u)
            dims.append(size)
    return tuple(dims)


def _parse_entry(b):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for f, _, v in _pb_fields(b):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif f == 7:
            e["sliced"] = True
    return e


def _pb_varint_field(f, v):
    return _put_varint(f << 3) + _put_varint(v)


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _build_entry(dtype, shape, offset, size, crc):
    dims = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
    out = _pb_varint_field(1, dtype) + _pb_bytes_field(2, dims)
    if offset:
        out += _pb_varint_field(4, offset)
    out += _pb_varint_field(5, size) + _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


# ------------------------------------------------------------------ LevelDB table
def _read_block(buf, offset, size, verify):
    body = buf[offset:offset + size]
    kind = buf[offset + size]
    if verify:
        want = unmask_crc(struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0])
        if crc32c(buf[offset:offset + size + 1]) != want:
            raise ValueError("tensor-bundle index: block checksum mismatch at offset %d" % offset)
    if kind != 0:
        raise ValueError("tensor-bundle index: compressed table blocks (type %d) are not supported" % kind)
    return body


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _read_table(path, verify=True):
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a LevelDB-format table (bad magic)" % path)
    footer = buf[-48:]
    _, pos = _get_varint(footer, 0)              # metaindex handle (offset, size) — unused
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


def _write_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _shortest_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator: a short key k with start <= k < limit."""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d < n and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _short_successor(key):
    """leveldb BytewiseComparator::FindShortSuccessor: a short key k >= key."""
    for i, c in enumerate(key):
        if c != 0xFF:
            return key[:i] + bytes([c + 1])
    return key


def _table_bytes(entries, block_size=262144):
    """The table as TensorFlow's io::TableBuilder (a copy of leveldb's) lays it out with BundleWriter's options
    (tensor_bundle.cc: kNoCompression; table options default block_size 256 KiB, restart interval 16): a data block is
    closed once its estimated size reaches block_size; its index key is the shortest separator to the next block's first
    key (the short successor for the last block); index block with restart interval 1; empty metaindex block; 48-byte
    footer.  A checkpoint of this model (about 200 keys) is one data block."""
    entries = sorted(entries)
    out = bytearray()

    def emit(body):
        off = len(out)
        out.extend(body)
        out.append(0)                                                   # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(body + b"\x00"))))
        return _put_varint(off) + _put_varint(len(body))

    index, cur, pending = [], [], None              # pending = (last key, handle) of a closed block awaiting its index key
    for k, v in entries:
        if pending is not None:
            index.append((_shortest_separator(pending[0], k), pending[1]))
            pending = None
        cur.append((k, v))
        body = _write_block(cur)
        if len(body) >= block_size:                 # BlockBuilder::CurrentSizeEstimate = entries + restart array + count
            pending = (k, emit(body))
            cur = []
    if cur:
        pending = (cur[-1][0], emit(_write_block(cur)))
    meta_handle = emit(_write_block([]))
    if pending is not None:
        index.append((_short_successor(pending[0]), pending[1]))
    index_handle = emit(_write_block(index, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out.extend(footer)
    return bytes(out)


def _write_table(path, entries, block_size=262144):
    with open(path, "wb") as f:
        f.write(_table_bytes(entries, block_size))


# ------------------------------------------------------------------ bundle
def latest_checkpoint(ckpt_dir):
    """tf.train.latest_checkpoint: the prefix named by <dir>/checkpoint, else the highest ckpt-N.index."""
    state = os.path.join(ckpt_dir, "checkpoint")
    if os.path.exists(state):
        with open(state) as f:
            m = re.search(r'^model_checkpoint_path:\s*"(.*)"', f.read(), re.M)
        if m:
            p = m.group(1)
            p = p if os.path.isabs(p) else os.path.join(ckpt_dir, p)
            if os.path.exists(p + ".index"):
                return p
    best = None
    if os.path.isdir(ckpt_dir):
        for name in os.listdir(ckpt_dir):
            m = re.match(r"^(.*-(\d+))\.index$", name)
            if m and (best is None or int(m.group(2)) > best[0]):
                best = (int(m.group(2)), os.path.join(ckpt_dir, m.group(1)))
    return best[1] if best else None


def read_bundle(prefix, verify=True, strip=True):
    """All numeric tensors of the bundle at `prefix` -> {variable path: ndarray}.  String tensors (the
    serialized object graph) and optimizer slot variables are skipped by the caller's key filter, not here.
    strip=False keeps the checkpoint keys as stored (with "/.ATTRIBUTES/VARIABLE_VALUE")."""
    entries = _read_table(prefix + ".index", verify)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index: missing bundle header" % prefix)
    num_shards, little = 1, True
    for f, _, v in _pb_fields(entries[0][1]):
        if f == 1:
            num_shards = v
        elif f == 2:
            little = v == 0
    if not little:
        raise ValueError("big-endian tensor bundles are not supported")
    shards = {}
    out = {}
    for key, val in entries[1:]:
        e = _parse_entry(val)
        name = key.decode()
        if e["dtype"] == _DT_STRING or e["sliced"]:
            continue
        if e["dtype"] not in _DTYPES:
            raise ValueError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64))
        if raw.size != count * dt.itemsize:
            raise ValueError("%s: %d bytes on disk for shape %s %s" % (name, raw.size, e["shape"], dt))
        if verify and e["crc32c"] is not None and crc32c(raw) != unmask_crc(e["crc32c"]):
            raise ValueError("%s: tensor checksum mismatch" % name)
        if strip and name.endswith(_SUFFIX):
            name = name[:-len(_SUFFIX)]
        out[name] = raw.view(dt).reshape(e["shape"]).copy()
    return out


_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"


def object_graph(paths):
    """Serialized CheckpointableObjectGraph for variables at `paths` ("a/b/kernel", ...; a dict {attribute path: tensor
    name} stores a variable under another checkpoint key than its path): see the module docstring.
    Nodes are numbered in breadth-first order from the root with the children of a node in name order, the way
    TensorFlow numbers them (tf.train.Checkpoint sorts its keyword arguments; checkpointable/util.py walks breadth first)."""
    names = dict(paths) if isinstance(paths, dict) else {p_: p_ for p_ in paths}
    tree = {}
    for path in sorted(names):
        cur = tree
        for part in path.split("/"):
            cur = cur.setdefault(part, {})
    nodes, queue = [{"children": [], "path": None}], [(0, tree, "")]
    while queue:
        nid, sub, prefix = queue.pop(0)
        for name in sorted(sub):
            cid = len(nodes)
            path = prefix + name
            nodes.append({"children": [], "path": path if not sub[name] else None})
            nodes[nid]["children"].append((name, cid))
            queue.append((cid, sub[name], path + "/"))
    out = b""
    for n in nodes:
        body = b""
        for name, nid in n["children"]:                          # ObjectReference {node_id = 1, local_name = 2}
            body += _pb_bytes_field(1, _pb_varint_field(1, nid) + _pb_bytes_field(2, name.encode()))
        if n["path"] is not None:                                # SerializedTensor {name = 1, full_name = 2, checkpoint_key = 3}
            body += _pb_bytes_field(2, _pb_bytes_field(1, b"VARIABLE_VALUE") + _pb_bytes_field(2, n["path"].encode())
                                    + _pb_bytes_field(3, (names[n["path"]] + _SUFFIX).encode()))
        out += _pb_bytes_field(1, body)
    return out


def graph_variables(nodes, max_depth=12):
    """Every way the object graph names a variable: {attribute path from the root: checkpoint key}.  restore() matches
    live objects to nodes edge by edge, so one node may be reachable along several paths (a Layer tracks a variable under
    its add_variable name AND through list attributes: estimator/matrix_0 and estimator/_matrices/0 are the same node);
    all of them are listed.  Cycles are cut (a node is not revisited along one path)."""
    out = {}

    def walk(nid, path, on_path):
        if not 0 <= nid < len(nodes) or nid in on_path or len(path) > max_depth:
            return
        for name, _full, key in nodes[nid]["attributes"]:
            if name == "VARIABLE_VALUE" and key:
                out.setdefault("/".join(path), key)
        for child, cid in nodes[nid]["children"].items():
            walk(cid, path + [child], on_path | {nid})
    walk(0, [], frozenset())
    return out


def parse_object_graph(blob):
    """-> list of nodes {"children": {local_name: node_id}, "attributes": [(name, full_name, checkpoint_key)]}"""
    nodes = []
    for f, _, v in _pb_fields(blob):
        if f != 1:
            continue
        node = {"children": {}, "attributes": []}
        for g, _, u in _pb_fields(v):
            if g == 1:
                d = {h: w for h, _, w in _pb_fields(u)}
                node["children"][bytes(d.get(2, b"")).decode()] = int(d.get(1, 0))
            elif g == 2:
                d = {h: bytes(w).decode() for h, _, w in _pb_fields(u) if h in (1, 2, 3)}
                node["attributes"].append((d.get(1, ""), d.get(2, ""), d.get(3, "")))
        nodes.append(node)
    return nodes


def _string_scalar_bytes(value):
    """On-disk form of a DT_STRING scalar (tensor_bundle.cc WriteStringTensor): [varint64 length][4-byte masked CRC-32C of
    the uint64 length][bytes]; returns (raw, entry crc) with the entry crc running over lengths, length checksum, bytes."""
    crc = crc32c(struct.pack("<Q", len(value)))
    length_ck = struct.pack("<I", mask_crc(crc))
    crc = crc32c(length_ck, crc)
    crc = crc32c(value, crc)
    return _put_varint(len(value)) + length_ck + value, mask_crc(crc)


def read_string_scalar(prefix, key=_GRAPH_KEY):
    """The bytes of a DT_STRING scalar entry of the bundle (None when the key is absent)."""
    for k, val in _read_table(prefix + ".index")[1:]:
        if k.decode() == key:
            e = _parse_entry(val)
            with open("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], 1), "rb") as f:
                f.seek(e["offset"])
                raw = f.read(e["size"])
            n, pos = _get_varint(raw, 0)
            return raw[pos + 4:pos + 4 + n]
    return None


def write_bundle(prefix, tensors, object_based=True, graph_paths=None):
    """Write {variable path: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001 (one shard).  Object-based
    (default): keys carry the "/.ATTRIBUTES/VARIABLE_VALUE" suffix and the serialized object graph of the variable
    paths is stored under "_CHECKPOINTABLE_OBJECT_GRAPH", which is what tf.train.Checkpoint.restore() walks.
    graph_paths = {attribute path: tensor name} writes a graph whose edges differ from the tensor names (tests)."""
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    entries = [(b"", _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1)))]     # num_shards=1, producer=1
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            a = a if a.flags.c_contiguous else np.ascontiguousarray(a)      # (ascontiguousarray would turn a scalar into [1])
            if a.dtype not in _DT_OF:
                raise ValueError("%s: dtype %s cannot be stored" % (name, a.dtype))
            raw = a.tobytes()
            f.write(raw)
            key = (name + _SUFFIX if object_based else name).encode()
            entries.append((key, _build_entry(_DT_OF[a.dtype], a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
        if object_based:
            graph = object_graph(graph_paths if graph_paths is not None else [n for n in tensors if ".OPTIMIZER_SLOT" not in n])
            raw, crc = _string_scalar_bytes(graph)
            f.write(raw)
            entries.append((_GRAPH_KEY.encode(), _build_entry(_DT_STRING, (), offset, len(raw), crc)))
    _write_table(prefix + ".index", entries)


def save_checkpoint(ckpt_dir, step, tensors):
    """checkpoint.save(file_prefix=ckpt_dir/ckpt) (train_hyper.py:255-268): writes ckpt-<step> and updates the
    `checkpoint` state file."""
    name = "ckpt-%d" % step
    write_bundle(os.path.join(ckpt_dir, name), tensors)
    state = os.path.join(ckpt_dir, "checkpoint")
    older = []
    if os.path.exists(state):
        with open(state) as f:
            older = re.findall(r'^all_model_checkpoint_paths:\s*"(.*)"', f.read(), re.M)
    paths = [p for p in older if p != name] + [name]
    with open(state, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for p in paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    return os.path.join(ckpt_dir, name)
