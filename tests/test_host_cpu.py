"""CPU-side checks (-m "not gpu"): the C-ABI libraries load and export every symbol include/pcgc.h
declares; the host tail (range coder, CDF quantiser, partition, ply text, container bytes) agrees
with the oracle and with the golden vectors produced by the reference's numpy code."""
import os
import re

import numpy as np
import pytest

from oracle import coder as ocoder
from pcgcv1_amd import _lib, coder_ops
from pcgcv1_amd.dataprocess import inout_bitstream as bs
from pcgcv1_amd.dataprocess import inout_points as iop

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libraries_export_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pcgc.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pcgc_[a-z0-9_]+)\s*\(", header))
    bound = set(_lib.HIP_API) | set(_lib.HOST_API)
    assert declared == bound, (declared ^ bound)
    hip, host = _lib.hip(), _lib.host()                     # raises if a symbol is missing
    for name in _lib.HIP_API:
        assert hasattr(hip, name)
    for name in _lib.HOST_API:
        assert hasattr(host, name)
    assert hip.pcgc_version() >= 1


def test_no_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pcgcv1_amd.models import model_voxception as m
    with pytest.raises(_lib.PcgcError):
        m.AnalysisTransform().load_weights({})


def _rand_pmf(rng, n, peaky):
    p = rng.random(n) ** peaky
    p /= p.sum()
    return np.maximum(p, 1e-9).astype(np.float32)


@pytest.mark.parametrize("seed", range(4))
def test_host_cdf_and_coder_bit_exact_vs_oracle(seed):
    rng = np.random.default_rng(seed)
    rows, cols, n = 64, 4, int(rng.integers(2, 30))
    pmf = np.stack([_rand_pmf(rng, n, rng.choice([1, 5, 15])) * rng.uniform(0.5, 1.0) for _ in range(rows * cols)])
    cdf = coder_ops.pmf_to_quantized_cdf(pmf)
    assert np.array_equal(cdf, ocoder.pmf_to_quantized_cdf(pmf))
    cdf = cdf.reshape(rows, cols, n + 1)
    sym = rng.integers(0, n, (rows, cols)).astype(np.int16)
    s = coder_ops.range_encode(sym, cdf)
    assert s == ocoder.range_encode(sym, cdf)
    assert np.array_equal(coder_ops.range_decode(s, sym.shape, cdf), sym)
    # broadcast table (entropy_model.py:219)
    cdf1 = cdf[:1]
    s = coder_ops.range_encode(sym, cdf1)
    assert s == ocoder.range_encode(sym, cdf1)
    assert np.array_equal(coder_ops.range_decode(s, sym.shape, cdf1), sym)


def test_host_batch_coder_matches_single_stream():
    rng = np.random.default_rng(9)
    B, S, n = 5, 3000, 7
    host = _lib.host()
    pmf = np.stack([_rand_pmf(rng, n, 10) for _ in range(B * S)])
    cdf = coder_ops.pmf_to_quantized_cdf(pmf)                      # [B*S, n+1]
    sym = np.array([np.searchsorted(cdf[i], rng.integers(0, 65536), side="right") - 1 for i in range(B * S)], np.int16)
    lohi = (cdf[np.arange(B * S), sym].astype(np.uint32) | ((cdf[np.arange(B * S), sym + 1] - 1).astype(np.uint32) << 16))
    cap = S * 2 + 64
    out = np.empty((B, cap), np.uint8)
    lens = np.zeros(B, np.int64)
    _lib.check_host(host.pcgc_range_encode_lohi_batch(_lib.nptr(lohi), B, S, 16, _lib.nptr(out), cap, _lib.nptr(lens), 3))
    strings = [out[i, :lens[i]].tobytes() for i in range(B)]
    for i in range(B):
        ref = ocoder.range_encode(sym[i * S:(i + 1) * S].reshape(-1, 1), cdf[i * S:(i + 1) * S].reshape(S, 1, n + 1))
        assert strings[i] == ref
    lower = np.full((B * S, n + 2), 0xFFFF, np.uint16)            # ncols > n on purpose
    lower[:, :n] = cdf[:, :n].astype(np.uint16)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    blob = np.frombuffer(b"".join(strings), np.uint8)
    dec = np.empty(B * S, np.int16)
    nsym = np.full(B, n, np.int32)
    _lib.check_host(host.pcgc_range_decode_u16_batch(_lib.nptr(blob), _lib.nptr(offs), _lib.nptr(lens), B, S,
                                                     _lib.nptr(lower), n + 2, _lib.nptr(nsym), 16, _lib.nptr(dec), 2))
    assert np.array_equal(dec, sym)


@pytest.mark.parametrize("name", ["a", "b", "d"])
def test_partition_matches_reference_golden(golden, name):
    g = golden("partition.npz")
    cube, min_num, scale = g[name + "_args"]
    pts = g[name + "_points"]
    pos, spos, cop = iop.partition(pts, int(cube), int(min_num))
    assert np.array_equal(pos, g[name + "_cube_positions"])
    assert np.array_equal(spos, iop.ordered_positions(pos))
    occ_lens = g[name + "_occ_lens"]
    occ = np.split(g[name + "_occ_flat"], np.cumsum(occ_lens)[:-1])
    cs = int(cube)
    for b in range(len(pos)):
        loc = pts[cop == b] % cs
        flat = np.unique((loc[:, 0] * cs + loc[:, 1]) * cs + loc[:, 2])
        assert np.array_equal(flat, occ[b])


def test_partition_edge_cases():
    with pytest.raises(ValueError):
        iop.partition(np.array([[1, 2, 3]] * 5, np.int32), 64, 20)
    with pytest.raises(ValueError):
        iop.partition(np.zeros((0, 3), np.int32), 64, 1)
    pts = np.array([[0, 0, 0]] * 3 + [[65, 1, 1]] * 2 + [[-1, 0, 0]] * 4, np.int32)       # negative coordinate: floor division
    pos, spos, cop = iop.partition(pts, 64, 2)
    assert pos.tolist() == [[0, 0, 0], [1, 0, 0], [-1, 0, 0]]
    assert sorted(np.bincount(cop).tolist()) == [2, 3, 4]


def test_ply_text_matches_reference_writer(golden, tmp_path):
    g = golden("partition.npz")
    for name in "abcd":
        assert iop.ply_bytes(g[name + "_points"]) == g[name + "_ply"].tobytes()
        f = tmp_path / ("%s.ply" % name)
        f.write_bytes(g[name + "_ply"].tobytes())
        assert np.array_equal(iop.load_ply_data(str(f)), g[name + "_points"])
    s = golden("select.npz")
    assert iop.ply_bytes(s["float_points"]) == s["float_ply"].tobytes()
    big = (np.arange(300, dtype=np.float32).reshape(100, 3) * np.float32(1 / 0.375))
    from oracle import points as op
    assert iop.ply_bytes(big) == op.ply_text(big).encode()
    rng = np.random.default_rng(5)                                          # enough points for several formatter blocks
    many = rng.integers(-70000, 70000, size=(100003, 3)).astype(np.int32)
    text = iop.ply_bytes(many)
    assert text == op.ply_text(many).encode()
    f = tmp_path / "many.ply"
    f.write_bytes(text)
    assert np.array_equal(iop.load_ply_data(str(f)), many)


def test_container_bytes_match_reference_writer(golden, tmp_path):
    g = golden("bitstream_hyper.npz")
    lens = g["y_lens"]
    cat = g["y_concat"].tobytes()
    ys, p = [], 0
    for l in lens:
        ys.append(cat[p:p + int(l)])
        p += int(l)
    sizes = bs.write_binary_files_hyper("g", ys, g["z_string"].tobytes(), g["points_numbers"], g["cube_positions"],
                                        g["y_min_vs"], g["y_max_vs"], g["y_shape"], int(g["z_min_v"]), int(g["z_max_v"]),
                                        g["z_shape"], rootdir=str(tmp_path), verbose=False)
    for ext in ("strings", "strings_head", "strings_hyper", "pointnums"):
        assert (tmp_path / ("g." + ext)).read_bytes() == g[ext].tobytes()
    assert sizes[:4] == tuple(int(v) for v in g["sizes"][:4])
    r = bs.read_binary_files_hyper("g", rootdir=str(tmp_path))
    assert r[0] == ys and r[1] == g["z_string"].tobytes()
    assert np.array_equal(r[2], g["points_numbers"])
    assert np.array_equal(np.unique(r[3], axis=0), np.unique(g["cube_positions"], axis=0))
    assert np.array_equal(np.unique(r[3], axis=0), np.unique(g["cubepos_decoded"], axis=0))     # same set tmc3 returns
    assert np.array_equal(r[4], g["y_min_vs"]) and np.array_equal(r[5], g["y_max_vs"])
    assert np.array_equal(r[6], g["y_shape"]) and (r[7], r[8]) == (-6, 5) and np.array_equal(r[9], g["z_shape"])
    with pytest.raises(ValueError):
        bs.pack_strings_head([b"x"], [-16], [3], g["y_shape"])
    with pytest.raises(ValueError):
        bs.pack_strings_head([b""], [-1], [3], g["y_shape"])


def test_crc32c_known_answers():
    """RFC 3720 B.4 vectors for CRC-32C (the tensor-bundle checksum)."""
    from pcgcv1_amd import tf_bundle
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283
    assert tf_bundle.crc32c(bytes(32)) == 0x8A9136AA
    assert tf_bundle.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tf_bundle.crc32c(bytes(range(32))) == 0x46DD794E
    assert tf_bundle.crc32c(b"6789", tf_bundle.crc32c(b"12345")) == 0xE3069283          # continuation
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert tf_bundle.unmask_crc(tf_bundle.mask_crc(c)) == c


def test_tensor_bundle_roundtrip_and_layout(tmp_path):
    """write -> read of a full model; structural checks of the LevelDB table and the bundle protos."""
    import struct
    from pcgcv1_amd import checkpoint, synthetic, tf_bundle
    w = synthetic.make_weights(seed=11)
    d = str(tmp_path / "a0.75b3")
    prefix = checkpoint.save_tf(w, d, 5000)
    assert os.path.basename(prefix) == "ckpt-5000"
    assert open(os.path.join(d, "checkpoint")).read().splitlines()[0] == 'model_checkpoint_path: "ckpt-5000"'
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xDB4775248B80FB57                          # leveldb table magic
    graph = tf_bundle.read_string_scalar(prefix)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == 4 * sum(v.size for v in w.values()) + len(graph) + 4 + len(tf_bundle._put_varint(len(graph)))
    assert b"analysis_transform/conv_in/bias/.ATTRIBUTES/VARIABLE_VALUE" in idx       # first key, not prefix-compressed
    got = checkpoint.load(d)
    assert sorted(got) == sorted(w)
    for k in w:
        assert got[k].dtype == np.float32 and got[k].shape == w[k].shape and np.array_equal(got[k], w[k]), k
    # a later step becomes the latest; optimizer slots / global_step / foreign keys are ignored by the loader
    w2 = {k: v + 1 for k, v in w.items()}
    extra = dict(w2)
    extra["global_step"] = np.asarray(7000, np.int64)
    extra["analysis_transform/conv_in/kernel/.OPTIMIZER_SLOT/main_optimizer/m"] = np.zeros((3, 3, 3, 1, 16), np.float32)
    tf_bundle.save_checkpoint(d, 7000, extra)
    assert tf_bundle.latest_checkpoint(d).endswith("ckpt-7000")
    checkpoint._CACHE.clear()
    got2 = checkpoint.load(d)
    assert sorted(got2) == sorted(w) and np.array_equal(got2["estimator/bais_0"], w["estimator/bais_0"] + 1)
    assert int(tf_bundle.read_bundle(prefix[:-4] + "7000")["global_step"]) == 7000
    # corruption is detected
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[100] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tf_bundle.read_bundle(prefix)


def test_checkpoint_object_graph_structure(tmp_path):
    """What tf.train.Checkpoint.restore walks (transform.py:107-112, train_hyper.py:107-121): the serialized object graph
    saved next to the tensors names every variable of models/spec.py exactly once, through attribute-name edges from
    the root, with the checkpoint key of its tensor; dtype / shape protos of every entry parse."""
    from pcgcv1_amd import checkpoint, synthetic, tf_bundle
    from pcgcv1_amd.models import spec
    w = synthetic.make_weights(seed=3)
    extra = dict(w)
    extra["global_step"] = np.asarray(12, np.int64)
    extra["hyper_encoder/conv1/kernel/.OPTIMIZER_SLOT/main_optimizer/m"] = np.zeros((3, 3, 3, 16, 16), np.float32)
    prefix = tf_bundle.save_checkpoint(str(tmp_path / "ck"), 12, extra)
    nodes = tf_bundle.parse_object_graph(tf_bundle.read_string_scalar(prefix))
    # root edges = the keyword names of the reference's tf.train.Checkpoint(...)
    assert sorted(nodes[0]["children"]) == ["analysis_transform", "estimator", "global_step", "hyper_decoder", "hyper_encoder",
                                            "synthesis_transform"] and nodes[0]["attributes"] == []
    keys, seen_nodes = {}, set()

    def walk(nid, path):
        assert nid not in seen_nodes and 0 <= nid < len(nodes)                       # a tree: every node reached once
        seen_nodes.add(nid)
        for name, full, key in nodes[nid]["attributes"]:
            assert name == "VARIABLE_VALUE" and full == "/".join(path) and key == full + "/.ATTRIBUTES/VARIABLE_VALUE"
            assert key not in keys
            keys[key] = nid
        for child, cid in nodes[nid]["children"].items():
            walk(cid, path + [child])
    walk(0, [])
    assert len(seen_nodes) == len(nodes)
    expected = set()
    for net, layers in spec.NETS.items():
        for l in layers():
            expected.add("%s/%s/kernel" % (net, l.name))
            if l.bias:
                expected.add("%s/%s/bias" % (net, l.name))
    expected |= {"estimator/%s_%d" % (k, i) for i in range(4) for k in ("matrix", "bais", "factor")} | {"global_step"}
    assert {k[:-len("/.ATTRIBUTES/VARIABLE_VALUE")] for k in keys} == expected == set(w) | {"global_step"}
    # nested Keras attribute edges, e.g. analysis_transform -> vrn1_1 -> conv1_1 -> kernel
    n = nodes[0]["children"]["analysis_transform"]
    for part in ("vrn1_1", "conv1_1", "kernel"):
        n = nodes[n]["children"][part]
    assert nodes[n]["attributes"][0][2] == "analysis_transform/vrn1_1/conv1_1/kernel/.ATTRIBUTES/VARIABLE_VALUE"
    # every index entry parses: dtype enum + dims (the graph itself is the one DT_STRING scalar)
    entries = tf_bundle._read_table(prefix + ".index")
    kinds = {}
    for k, v in entries[1:]:
        e = tf_bundle._parse_entry(v)
        kinds[k.decode()] = (e["dtype"], e["shape"])
    assert kinds["_CHECKPOINTABLE_OBJECT_GRAPH"] == (7, ())
    assert kinds["global_step/.ATTRIBUTES/VARIABLE_VALUE"] == (9, ())
    assert kinds["analysis_transform/conv_in/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == (1, (3, 3, 3, 1, 16))
    assert all(k in kinds for k in keys)
    # and the bundle still loads here (string tensor skipped, slot ignored)
    got = checkpoint.load(str(tmp_path / "ck"))
    assert sorted(got) == sorted(w)


def test_hand_assembled_bundle(tmp_path):
    """tests/golden/tf_bundle_min.* was assembled byte by byte from the published table / proto formats by
    tools/make_tf_bundle_fixture.py (its own varints, CRC-32C, block and footer layout; it does not import tf_bundle):
    the reader parses it and the writer reproduces both files byte for byte."""
    from pcgcv1_amd import tf_bundle
    g = os.path.join(ROOT, "tests", "golden", "tf_bundle_min")
    got = tf_bundle.read_bundle(g)
    assert sorted(got) == ["estimator/matrix_0", "global_step"]
    assert got["estimator/matrix_0"].dtype == np.float32 and got["estimator/matrix_0"].shape == (2, 3, 1)
    assert got["estimator/matrix_0"].reshape(-1).tolist() == [0.5, -1.25, 2.0, 0.0, 3.75, -8.0]
    assert got["global_step"].dtype == np.int64 and got["global_step"].shape == () and int(got["global_step"]) == 5000
    nodes = tf_bundle.parse_object_graph(tf_bundle.read_string_scalar(g))
    assert tf_bundle.graph_variables(nodes) == {"estimator/matrix_0": "estimator/matrix_0/.ATTRIBUTES/VARIABLE_VALUE",
                                                "global_step": "global_step/.ATTRIBUTES/VARIABLE_VALUE"}
    p = str(tmp_path / "again")
    tf_bundle.write_bundle(p, got)
    for ext in (".index", ".data-00000-of-00001"):
        assert open(p + ext, "rb").read() == open(g + ext, "rb").read(), ext
    # the generator is deterministic and committed: running it again gives the committed bytes
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tools", "make_tf_bundle_fixture.py"))
    mk = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mk)
    mk.OUT = str(tmp_path / "regen")
    mk.main()
    for ext in (".index", ".data-00000-of-00001"):
        assert open(mk.OUT + ext, "rb").read() == open(g + ext, "rb").read(), ext


def test_checkpoint_binding_walks_the_object_graph(tmp_path):
    """checkpoint.load binds variables the way tf.train.Checkpoint.restore does (transform.py:107-112): through the file's
    object graph from the root edges, not by key spelling.  (1) tensors stored under opaque keys bind through the graph;
    (2) the entropy bottleneck's variables reachable only through its list attributes (entropy_model.py:47-66:
    estimator/_matrices/0 ...) bind through the alias edges; (3) a bundle without a graph binds by key name; (4) every
    expected variable that cannot be bound is named in the error."""
    from pcgcv1_amd import checkpoint, synthetic, tf_bundle
    w = synthetic.make_weights(seed=5)
    names = sorted(w)
    # (1) opaque checkpoint keys, plain attribute-path graph
    opaque = {"v/%03d" % i: w[n] for i, n in enumerate(names)}
    d1 = str(tmp_path / "opaque")
    tf_bundle.write_bundle(os.path.join(d1, "ckpt-1"), opaque, graph_paths={n: "v/%03d" % i for i, n in enumerate(names)})
    got = checkpoint.load(d1)
    assert sorted(got) == names and all(np.array_equal(got[n], w[n]) for n in names)
    assert all(v.startswith("graph:") for v in checkpoint.LAST_BINDING.values())
    # (2) estimator variables only under the list attributes
    lists = {"matrix": "_matrices", "bais": "_biases", "factor": "_factors"}

    def alias(n):
        if n.startswith("estimator/"):
            kind, i = n[len("estimator/"):].rsplit("_", 1)
            return "estimator/%s/%s" % (lists[kind], i)
        return n
    d2 = str(tmp_path / "alias")
    tf_bundle.write_bundle(os.path.join(d2, "ckpt-1"), {alias(n): w[n] for n in names})
    got = checkpoint.load(d2)
    assert sorted(got) == names and np.array_equal(got["estimator/bais_2"], w["estimator/bais_2"])
    assert checkpoint.LAST_BINDING["estimator/bais_2"] == "graph:estimator/_biases/2"
    # (3) name-based bundle (tf.train.Saver style: no suffix, no graph)
    d3 = str(tmp_path / "saver")
    tf_bundle.write_bundle(os.path.join(d3, "ckpt-1"), w, object_based=False)
    got = checkpoint.load(d3)
    assert sorted(got) == names and checkpoint.LAST_BINDING["analysis_transform/conv_in/kernel"] == "key:analysis_transform/conv_in/kernel"
    # (4) unbound variables are reported by name
    broken = {n: v for n, v in w.items() if n not in ("hyper_decoder/conv4_2/bias", "synthesis_transform/up_1/kernel")}
    d4 = str(tmp_path / "broken")
    tf_bundle.write_bundle(os.path.join(d4, "ckpt-1"), broken)
    with pytest.raises(ValueError, match="2 of .* expected variables.*hyper_decoder/conv4_2/bias"):
        checkpoint.load(d4)
    # a factorized checkpoint (no hyperprior nets, 16-channel estimator) is complete as it is
    fact = {n: v for n, v in w.items() if not n.startswith("hyper_")}
    d5 = str(tmp_path / "fact")
    tf_bundle.write_bundle(os.path.join(d5, "ckpt-1"), fact)
    assert sorted(checkpoint.load(d5)) == sorted(fact)


def test_object_code_has_no_unprotected_128_bit_stores():
    """tools/check_isa.py on the built objects: the gfx950 store-data hazard the row kernels met in round 2 cannot be
    reintroduced by a compiler bump or an edit without this test noticing (no 128-bit buffer store with a register soffset —
    the form LLVM's hazard recognizer does not protect)."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    ci = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(ci)
    if not os.path.exists(os.path.join(ci.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not found")
    objs = [os.path.join(ci.OBJ, f) for f in sorted(os.listdir(ci.OBJ)) if f.endswith(".hip.o")]
    assert len(objs) >= 10, "run `python -m pcgcv1_amd.build` first"
    stores = mfmas = 0
    for obj in objs:
        n_mfma, n_store, bad, _ = ci.check_text(ci.disassemble(obj))
        assert bad == [], bad[:5]
        stores += n_store
        mfmas += n_mfma
    assert stores > 100 and mfmas > 10000            # the disassembly really covered the row kernels
    # the rule itself, on two lines of disassembly
    ok = "\tbuffer_store_dwordx4 v[152:155], v172, s[4:7], 0 offen     // 00C820: E07C1000"
    hazard = "\tbuffer_store_dwordx4 v[152:155], v172, s[4:7], s12 offen   // 00C820: E07C1000"
    assert ci.check_text(ok)[2] == [] and len(ci.check_text(hazard)[2]) == 1


def test_h5_training_cube_reader(tmp_path):
    """The reference's training files are HDF5 (generate_dataset.py:27-29: dataset 'data', uint8 [n,3]); h5py is not in this
    image, so dataprocess/h5min.py reads them.  Pinned to tests/golden/cube_points.h5, which tools/make_h5_fixture.py
    assembles byte by byte from the HDF5 format specification without importing the reader; unsupported features are
    refused by name, never mis-read."""
    import importlib.util
    from pcgcv1_amd.dataprocess import h5min
    spec_ = importlib.util.spec_from_file_location("mkh5", os.path.join(ROOT, "tools", "make_h5_fixture.py"))
    mk = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mk)
    path = os.path.join(ROOT, "tests", "golden", "cube_points.h5")
    a = h5min.read_dataset(path, "data")
    assert a.dtype == np.uint8 and a.shape == (37, 3) and np.array_equal(a, np.array(mk.points(), np.uint8))
    mk.OUT = str(tmp_path / "again.h5")
    mk.main()
    assert open(mk.OUT, "rb").read() == open(path, "rb").read()             # the committed fixture is what the script writes
    with pytest.raises(KeyError, match="no dataset 'points'"):
        h5min.read_dataset(path, "points")
    raw = bytearray(open(path, "rb").read())
    raw[8] = 2                                                              # a libver='latest' superblock
    (tmp_path / "v2.h5").write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        h5min.read_dataset(str(tmp_path / "v2.h5"))
    # the training driver's loader takes the file (h5py when installed, the minimal reader here)
    pytest.importorskip("torch")
    import sys
    if "h5py" not in sys.modules:
        from pcgcv1_amd.train_hyper import load_cube_points
        pts = load_cube_points(path)
        assert pts.dtype == np.int64 and np.array_equal(pts, a.astype(np.int64))


def test_ply_parser_follows_python_float_rules(tmp_path):
    """pcgc_parse_ply_points against the reference's line rule (split(' '), float() of the first three tokens,
    ValueError -> skip the line; inout_points.py:15-22) on awkward lines, and on a large seeded cloud."""
    txt = ("ply\nformat ascii 1.0\nelement vertex 9\nproperty float x\nend_header\n1 2 3\n4.5 -6.25 7e2 9 9\n1  2 3\n 8 9 10\n"
           "11 12 13 \n1\t2\t3\n+1.0 .5 5.\n-0 1 2\nnan 1 2\n0x10 1 2\n1_0 2 3\n1__0 2 3\n7 8\n 1 2\n123456789012345678 1 1\n3 4 5")
    f = tmp_path / "odd.ply"
    f.write_text(txt)
    ref = []
    for line in txt.split("\n"):
        w = (line + "\n").split(" ")
        try:
            ref.append([float(w[0]), float(w[1]), float(w[2])])
        except (ValueError, IndexError):
            continue
    with np.errstate(invalid="ignore"):
        ref = np.array(ref).astype(np.int32)
    assert np.array_equal(iop.load_ply_data(str(f)), ref)
    rng = np.random.default_rng(4)
    pts = rng.integers(-5000, 70000, (200_000, 3)).astype(np.int32)
    g = tmp_path / "big.ply"
    iop.write_ply_data(str(g), pts)
    assert np.array_equal(iop.load_ply_data(str(g)), pts)
    fl = (pts[:5000].astype(np.float32) * np.float32(1 / 0.375))
    iop.write_ply_data(str(g), fl)                                   # float text, e.g. '26666.666'
    assert np.array_equal(iop.load_ply_data(str(g)), fl.astype(np.float64).astype(np.int32))


def test_generate_dataset_writes_the_partition_cubes(tmp_path):
    """generate_dataset.py:11-38: one uint8 [n,3] file per cube of >= 20 points, named <stem>_<i>n, holding exactly the
    in-cube coordinates load_points returns; the 1/9 hold-out split of train_hyper.py:167, 257."""
    from pcgcv1_amd import generate_dataset, synthetic, train_hyper
    from pcgcv1_amd.dataprocess import inout_points as iop
    (tmp_path / "in").mkdir()
    pts = synthetic.make_cloud(seed=4, res=128, n_shells=3, rmin=0.2, rmax=0.45)
    iop.write_ply_data(str(tmp_path / "in" / "a.ply"), pts)
    files = generate_dataset.generate_dataset(str(tmp_path / "in"), str(tmp_path / "out"), 1e6, cube_size=32, seed=0)
    set_points, _ = iop.load_points(str(tmp_path / "in" / "a.ply"), cube_size=32, min_num=20)
    assert len(files) == len(set_points) and all(f.endswith("n.npy") for f in files)
    got = sorted(tuple(map(tuple, train_hyper.load_cube_points(f))) for f in files)
    want = sorted(tuple(map(tuple, np.asarray(p, np.int64))) for p in set_points)
    assert got == want
    assert np.load(files[0]).dtype == np.uint8
    held, train = train_hyper.split_file_list(list(range(20)))
    assert held == [0, 1] and train == list(range(2, 20))
    # the reference's container (HDF5 dataset 'data', generate_dataset.py:27-29): written and read without h5py here
    files5 = generate_dataset.generate_dataset(str(tmp_path / "in"), str(tmp_path / "out5"), 1e6, cube_size=32, fmt="h5", seed=5)
    assert len(files5) == len(files) and all(f.endswith("n.h5") for f in files5)
    got5 = sorted(tuple(map(tuple, train_hyper.load_cube_points(f))) for f in files5)
    assert got5 == want
    with pytest.raises(ValueError):
        generate_dataset.generate_dataset(str(tmp_path / "in"), str(tmp_path / "o2"), 1, cube_size=512)


def test_entropy_slices_cover_the_batch_on_launch_boundaries():
    """conditional_entropy_model._slices: contiguous, complete, every boundary but the last on a multiple of 8 cubes (the
    64^3 stage's launch size), no slice under 32 cubes unless the batch is."""
    from pcgcv1_amd.models.conditional_entropy_model import _slices
    for B in (1, 7, 31, 32, 33, 63, 64, 65, 96, 102, 103, 205, 1000):
        for n in (1, 2, 3, 4):
            sl = _slices(B, n)
            assert sl[0][0] == 0 and sl[-1][1] == B and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert all(hi > lo for lo, hi in sl) and len(sl) <= n
            assert all(hi % 8 == 0 for lo, hi in sl[:-1])
            if B >= 64 and len(sl) > 1:
                assert min(hi - lo for lo, hi in sl[:-1]) >= 32


def test_decoder_slices_by_cloud_size_and_row_width():
    """conditional_entropy_model.decode_slices: contiguous and complete; a 24-cube first slice (what the GPU waits for at the
    start of a decode) whenever 32 cubes are left for the rest; a pipeline of many hundred cubes keeps slices of about 100
    cubes (that wait must not grow with the cloud)."""
    from pcgcv1_amd.models.conditional_entropy_model import decode_slices
    for B in (1, 31, 46, 55, 56, 103, 205, 820, 1316, 2631):
        for row_bytes in (0, 10, 34):
            sl = decode_slices(B, row_bytes=row_bytes)
            assert sl[0][0] == 0 and sl[-1][1] == B and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert all(hi > lo for lo, hi in sl)
            assert all(hi % 8 == 0 for lo, hi in sl[:-1])
            if B >= 200:
                assert max(hi - lo for lo, hi in sl) <= 128
            assert sl[0] == ((0, 24) if B >= 56 else (0, B))
    assert decode_slices(103) == [(0, 24), (24, 103)]                    # the bench's pipelines
    # a caller that consumes the slices as they finish (process.StreamedPostprocess) gets a short LAST slice as well
    assert decode_slices(103, tail=24) == [(0, 24), (24, 72), (72, 103)] and decode_slices(102, tail=24) == [(0, 24), (24, 72), (72, 102)]
    assert decode_slices(60, tail=24) == decode_slices(60) and decode_slices(820, tail=24) == decode_slices(820)
    assert decode_slices(103, first=0, n=2) == [(0, 56), (56, 103)] and decode_slices(103, first=16, n=2)[0] == (0, 16)


def test_range_encode_values_equals_range_encode_of_shifted_symbols():
    """pcgc_range_encode_values (values int8 / int16 and an offset, the z string's path) against pcgc_range_encode of
    values - offset: the same bytes; a value outside the table is an error, not a wrapped symbol."""
    from pcgcv1_amd import _lib, coder_ops
    rng = np.random.default_rng(11)
    C, n = 8, 11
    pmf = rng.random((C, n)).astype(np.float32)
    pmf /= pmf.sum(1, keepdims=True)
    cdf = coder_ops.pmf_to_quantized_cdf(pmf, precision=16).reshape(1, C, -1)
    for rows in (0, 1, 777, 20000):
        vals = rng.integers(-5, 6, (rows, C)).astype(np.int16)
        want = coder_ops.range_encode((vals + 5).astype(np.int16), cdf)
        assert coder_ops.range_encode_values(vals, -5, cdf) == want
        assert coder_ops.range_encode_values(vals.astype(np.int8), -5, cdf) == want
        if rows:
            assert coder_ops.range_decode(want, (rows, C), cdf).tolist() == (vals + 5).tolist()
    with pytest.raises(_lib.PcgcError, match="outside"):
        coder_ops.range_encode_values(np.full((3, C), 6, np.int16), -5, cdf)


def _reference_rho_loop(rhos, psnr_of):
    """eval_ablation_studies.py:156-172 traced by hand on a ladder: i == 0 sets the maximum to 0 (not to the first PSNR),
    later steps update it, the loop breaks at the first PSNR below it."""
    best, mx = None, 0.0
    for i, rho in enumerate(rhos):
        p = psnr_of(rho)
        mx = 0.0 if i == 0 else max(p, mx)
        if p < mx:
            break
        best = rho
    return best


def test_select_optimal_rho_follows_the_reference_ladder_walk():
    from pcgcv1_amd import eval as rd
    assert rd.RHOS_D1[:4] == [0.8, 0.9, 1.0, 1.02] and rd.RHOS_D1[-1] == 3.0 and len(rd.RHOS_D1) == 16
    assert rd.RHOS_D2[:3] == [1.0, 0.98, 0.95] and rd.RHOS_D2[-1] == 0.30 and len(rd.RHOS_D2) == 15
    rng = np.random.default_rng(5)
    for t in range(300):
        rhos = rd.RHOS_D1 if t % 2 else rd.RHOS_D2
        peak = rng.integers(0, len(rhos))
        ps = {r: 70.0 - 0.3 * abs(i - peak) + (rng.random() * 0.5 if t % 3 == 0 else 0.0) for i, r in enumerate(rhos)}
        log = []
        got = rd.select_optimal_rho("k", rhos, lambda r: {"k": ps[r]}, log)
        assert got == _reference_rho_loop(rhos, ps.__getitem__)
        assert [l[2] for l in log] == rhos[:len(log)]                      # walked in ladder order, stopped early
    # the reference's quirk: the second entry always replaces the first, even when it is worse
    assert rd.select_optimal_rho("k", [0.8, 0.9, 1.0], lambda r: {"k": {0.8: 70.0, 0.9: 60.0, 1.0: 50.0}[r]}) == 0.9
    # unimodal curve: the maximum
    assert rd.select_optimal_rho("k", [1.0, 0.98, 0.95, 0.92], lambda r: {"k": {1.0: 70, 0.98: 71, 0.95: 72, 0.92: 71.5}[r]}) == 0.95


def test_rho_search_writes_the_ini_and_default_config_lists_the_seven_hyper_rates(tmp_path):
    import configparser
    from pcgcv1_amd import eval as rd
    ck = tmp_path / "ck"
    for name in ("a0.75b3.00", "a2.00b3.00", "a6b3"):
        (ck / name).mkdir(parents=True)
    cfg, path = rd.set_default_config(str(tmp_path / "longdress_vox10_1300.ply"), str(tmp_path / "cfg"), 1024, ckpt_root=str(ck))
    assert os.path.basename(path) == "longdress_vox10_1300.ini"
    assert cfg.sections() == ["R1", "R2", "R3", "R4", "R5", "R6", "R7"]                      # eval_ablation_studies.py:71-77
    assert [float(cfg.get(r, "scale")) for r in cfg.sections()] == [0.625, 1, 1, 1, 1, 1, 1]
    assert cfg.get("R1", "ckpt_dir") == cfg.get("R2", "ckpt_dir") == str(ck / "a0.75b3.00") + "/"
    assert cfg.get("R5", "ckpt_dir") == str(ck / "a6b3") + "/" and cfg.get("R7", "ckpt_dir") == str(ck / "a16b3") + "/"
    assert cfg.getint("DEFAULT", "cube_size") == 64 and cfg.getint("DEFAULT", "min_num") == 64 and cfg.getint("DEFAULT", "resolution") == 1024
    calls = []

    def measure(rho):
        calls.append(rho)
        return {"mseF,PSNR (p2point)": 70.0 - abs(rho - 1.1), "mseF,PSNR (p2plane)": 74.0 - abs(rho - 0.9)}
    assert rd.cfg_post_process(cfg, path, "R3", measure) == (1.1, 0.9)
    back = configparser.ConfigParser()
    back.read(path)
    assert float(back.get("R3", "rho_d1")) == 1.1 and float(back.get("R3", "rho_d2")) == 0.9 and not back.has_option("R4", "rho_d1")
    n = len(calls)
    assert rd.cfg_post_process(back, path, "R3", measure) == (1.1, 0.9) and len(calls) == n        # read, not searched again
    # no normals: no point-to-plane search, nothing written for rho_d2
    assert rd.cfg_post_process(back, path, "R4", measure, have_normals=False) == (1.1, 1.0)
    again, _ = rd.set_default_config(str(tmp_path / "longdress_vox10_1300.ply"), str(tmp_path / "cfg"), 512, ckpt_root=str(ck))
    assert again.has_option("R4", "rho_d1") and not again.has_option("R4", "rho_d2") and again.getint("DEFAULT", "resolution") == 1024


def test_bench_refuses_rccl_with_fewer_devices_than_ranks():
    """`bench.py --gpus N` over RCCL with fewer than N visible devices says so and exits non-zero at once — before any
    process group exists — instead of waiting in the first collective for the 300 s timeout (this container has no GPU:
    0 devices for 2 ranks)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    env.pop("PCGC_BENCH_BACKEND", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "needs 2 visible devices" in r.stderr and time.time() - t0 < 60


def test_bdrate_metrics_match_reference_golden():
    """myutils/bdrate_metrics.py (bdsnr 28-75, bdrate 78-129): tests/golden/bdrate_cases.json was made by running the
    reference's module in the build container on this repository's six-point RD table and on seeded curves."""
    import json
    import math
    from pcgcv1_amd.myutils import bdrate_metrics as bd
    with open(os.path.join(os.path.dirname(__file__), "golden", "bdrate_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 12
    for c in cases:
        c1, c2 = [tuple(p) for p in c["curve1"]], [tuple(p) for p in c["curve2"]]
        assert abs(bd.bdsnr(c1, c2) - c["bdsnr"]) <= 1e-9 * max(1.0, abs(c["bdsnr"]))
        assert abs(bd.bdrate(c1, c2) - c["bdrate"]) <= 1e-9 * max(1.0, abs(c["bdrate"]))
    same = [(0.1, 60.0), (0.2, 62.0), (0.3, 63.0), (0.4, 64.0)]
    assert abs(bd.bdsnr(same, same)) < 1e-9 and abs(bd.bdrate(same, same)) < 1e-9
    touching = [(0.1, 64.0), (0.2, 65.0), (0.3, 66.0), (0.4, 67.0)]
    assert math.isnan(bd.bdrate(same, touching))                      # 0 / 0 in the reference as well
    assert bd.bdsnr(same, [(0.4, 64.0), (0.5, 65.0), (0.6, 66.0), (0.7, 67.0)]) == 0.0
    assert bd.bdrate(same, [(r * 100.0, p) for r, p in same]) > 9000    # 100 x the rate: +9900 %


def test_binary_ply_input(tmp_path):
    """load_ply_data / load_ply_normals read binary_little_endian and binary_big_endian plys (an extension: the reference's
    loader, inout_points.py:8-28, parses text only): same points as the ASCII file of the same cloud, extra vertex
    properties skipped, normals picked up by name, trailing elements ignored."""
    from pcgcv1_amd.dataprocess import inout_points as iop
    rng = np.random.default_rng(5)
    pts = rng.integers(0, 1024, (1000, 3)).astype(np.int32)
    nrm = rng.standard_normal((1000, 3)).astype(np.float32)
    iop.write_ply_data(str(tmp_path / "a.ply"), pts)
    ascii_pts = iop.load_ply_data(str(tmp_path / "a.ply"))
    for order, fmt in (("<", "binary_little_endian"), (">", "binary_big_endian")):
        rec = np.zeros(1000, np.dtype([("x", order + "f4"), ("y", order + "f4"), ("z", order + "f4"), ("red", "u1"), ("nx", order + "f4"),
                                       ("ny", order + "f4"), ("nz", order + "f4")]))
        for i, k in enumerate("xyz"):
            rec[k] = pts[:, i]
            rec["n" + k] = nrm[:, i]
        head = ("ply\nformat %s 1.0\ncomment made by a test\nelement vertex 1000\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty float nx\nproperty float ny\nproperty float nz\nelement face 1\nproperty list uchar int vertex_indices\n"
                "end_header\n" % fmt).encode()
        path = tmp_path / (fmt + ".ply")
        path.write_bytes(head + rec.tobytes() + bytes([3]) + np.array([0, 1, 2], order + "i4").tobytes())
        assert np.array_equal(iop.load_ply_data(str(path)), ascii_pts)
        p2, n2 = iop.load_ply_normals(str(path))
        assert np.array_equal(p2, pts) and np.array_equal(n2, nrm)
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty list uchar int x\nend_header\n\x00")
    with pytest.raises(ValueError, match="not a scalar"):
        iop.load_ply_data(str(bad))
    # a header longer than the first 4 KiB (many comment lines) is still recognised as binary by BOTH entry points ...
    rec = np.zeros(1000, np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]))
    for i, k in enumerate("xyz"):
        rec[k] = pts[:, i]
    long_head = ("ply\nformat binary_little_endian 1.0\n" + "comment padding padding padding padding\n" * 300 +
                 "element vertex 1000\nproperty float x\nproperty float y\nproperty float z\nend_header\n").encode()
    assert len(long_head) > 8192
    (tmp_path / "long.ply").write_bytes(long_head + rec.tobytes())
    assert np.array_equal(iop.load_ply_data(str(tmp_path / "long.ply")), ascii_pts)
    assert np.array_equal(iop.load_ply_normals(str(tmp_path / "long.ply"))[0], pts)
    # ... and a `ply` file whose header never ends is refused, not parsed as text
    (tmp_path / "cut.ply").write_bytes(b"ply\nformat ascii 1.0\n" + b"comment x\n" * 1000)
    with pytest.raises(ValueError, match="end_header"):
        iop.load_ply_data(str(tmp_path / "cut.ply"))


def test_bench_traffic_record_matches_the_committed_profiles():
    """bench.py's `roofline.traffic` comes from the newest committed PMC summary (profiles/*pmc_per_kernel.csv): the row is chosen
    by launch type (dense launches of the synthesis / skipping launches of the analysis), its duration is checked against the plain
    kernel trace of the same collection, and a summary that does not describe the timed kernel is refused, not quoted.  Runs on a
    FIXTURE pair of summaries (tests/golden/profiles_fixture/: the vrn16a / vrn16bc rows of one committed collection) so that the
    next profile commit or a changed template signature cannot break the CPU suite; the launch types are told apart the way bench.py
    does it — by dispatch count — not by template-argument strings."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import csv
    fx = os.path.join(root, "tests", "golden", "profiles_fixture")
    rows = [r for r in csv.DictReader(open(os.path.join(fx, "fx_kernel_stats_pipes1.csv"))) if "vrn16a_row_kernel" in r["Name"]]
    assert len(rows) == 2
    rows.sort(key=lambda r: int(r["Calls"]))                     # fewer launches = the analysis' skipping launches (two chunks each)
    live = {"skip": float(rows[0]["AverageNs"]) / 1e6, "dense": float(rows[1]["AverageNs"]) / 1e6}
    names = {"skip": rows[0]["Name"].split("(")[0], "dense": rows[1]["Name"].split("(")[0]}
    dense, rec = bench._traffic_from_profiles("vrn16a_row_kernel@D64", live["dense"] * 1.02, profiles_dir=fx)
    assert dense is not None and "refused" not in rec, rec
    assert rec["kernel_row"] == names["dense"]
    assert abs(rec["duration_vs_live"]) < 0.10 and rec.get("trace_file", "") == "fx_kernel_stats_pipes1.csv"
    assert 2.4e8 < dense < 3.0e8                                                     # FETCH x 2 + WRITE of one 8-cube launch: 269 MB
    skip, rec_s = bench._traffic_from_profiles("vrn16a_row_kernel@D64 [analysis: empty tiles skipped]", live["skip"] * 0.98, profiles_dir=fx)
    assert skip is not None and rec_s["kernel_row"] == names["skip"] and skip != dense
    none, rec_bad = bench._traffic_from_profiles("vrn16a_row_kernel@D64", 2.0 * live["dense"], profiles_dir=fx)    # twice the duration: another kernel / launch size
    assert none is None and "refused" in rec_bad
    # the live directory still parses (whatever its newest collection is): a record or nothing, never an exception
    bench._traffic_from_profiles("vrn16a_row_kernel@D64", live["dense"])


def test_decoder_plan_covers_every_cube_once_and_starts_with_the_first_cubes():
    """transform._decode_plan: which cubes each decoder pipeline decodes, slice by slice.  Every cube exactly once; with
    interleaving the pipelines' short first slices are the FIRST cubes of the cloud in z order (the second pipeline waits for
    48 cubes' z symbols, not 127); without it (the CLI's streamed tail, PCGC_DEC_INTERLEAVE=0) the contiguous groups cut by
    decode_slices."""
    from pcgcv1_amd import transform
    for B in (205, 96, 97, 130, 1640, 48, 1):
        groups = transform._groups(B)
        for tail in (0, 24):
            for inter in (True, False):
                plan = transform._decode_plan(B, groups, tail=tail, interleave=inter)
                assert len(plan) == len(groups)
                flat = sorted(s for p in plan for s in p)
                assert flat[0][0] == 0 and flat[-1][1] == B and all(a[1] == b[0] for a, b in zip(flat, flat[1:])), (B, plan)
                assert all(b > a for a, b in flat)
                for p in plan:                               # a pipeline decodes its slices in z order
                    assert p == sorted(p)
                if not inter:
                    for (lo, hi), p in zip(groups, plan):
                        assert p[0][0] == lo and p[-1][1] == hi
    plan = transform._decode_plan(205, transform._groups(205), interleave=True)
    assert plan == [[(0, 24), (48, 127)], [(24, 48), (127, 205)]]

