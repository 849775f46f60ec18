"""-m gpu: SURVEY §8e with the real kernels — the sharded codec (sharding.HipOps) on world sizes 1, 2 and 3
(ranks share the test box's single GPU, rendezvous over gloo on 127.0.0.1) must reproduce the single-process
compress_hyper / decompress_hyper bitstream and reconstruction bit for bit (batch-slot invariance of the kernels
+ order-preserving contiguous blocks + the one z-range exchange)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
N_CUBES = 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp_path, backend="gloo", n_cubes=N_CUBES):
    out = str(tmp_path / ("w%d%s.pkl" % (world, backend)))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker_gpu.py"), str(r), str(world), str(port), out,
                               backend, str(n_cubes)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    with open(out, "rb") as f:
        return pickle.load(f)


def test_sharded_codec_equals_single_process(tmp_path):
    from pcgcv1_amd import synthetic, transform
    from pcgcv1_amd.dataprocess import inout_points as iop
    from pcgcv1_amd.models import model_voxception as model
    cubes = synthetic.make_cubes(seed=9, n_cubes=N_CUBES, cube_size=32, occupancy=0.03)
    nums = cubes.sum(axis=(1, 2, 3, 4)).astype(np.uint16)
    ref = transform.compress_hyper(cubes, model, "synthetic:21:dense")
    ref_logits = transform.decompress_hyper(*ref, model, "synthetic:21:dense")
    ref_masks = iop.select_voxels(ref_logits, nums, 1.0).cpu().numpy()
    ref_logits = ref_logits.cpu().numpy()
    # (1, "nccl"): a one-rank RCCL group — the box has one GPU, but every collective of the path (all_reduce,
    # all_gather_into_tensor, broadcast on HBM buffers) really runs through RCCL
    for world, backend in ((1, "gloo"), (2, "gloo"), (3, "gloo"), (1, "nccl")):
        got = _run(world, tmp_path, backend)
        s = got["stream"]
        assert list(s[0]) == list(ref[0]) and s[4] == ref[4], world          # y strings in cube order, the single z string
        for i in (1, 2, 3, 7):
            assert np.array_equal(np.asarray(s[i]), np.asarray(ref[i])), (world, i)
        assert (s[5], s[6]) == (ref[5], ref[6])
        assert np.array_equal(got["logits"], ref_logits), world
        assert np.array_equal(got["masks"], ref_masks.astype(np.uint8)), world


@pytest.mark.parametrize("n_cubes", [1, 2, 4])
def test_sharded_codec_with_fewer_cubes_than_ranks(tmp_path, n_cubes):
    """Three ranks, one / two / four cubes: ranks whose encoder block or (geometrically shrinking) decoder block is empty take
    part in every collective with zero-size payloads; same bytes and logits as one process."""
    from pcgcv1_amd import synthetic, transform
    from pcgcv1_amd.models import model_voxception as model
    cubes = synthetic.make_cubes(seed=9, n_cubes=n_cubes, cube_size=32, occupancy=0.03)
    ref = transform.compress_hyper(cubes, model, "synthetic:21:dense")
    ref_logits = transform.decompress_hyper(*ref, model, "synthetic:21:dense").cpu().numpy()
    got = _run(3, tmp_path, "gloo", n_cubes=n_cubes)
    s = got["stream"]
    assert list(s[0]) == list(ref[0]) and s[4] == ref[4]
    for i in (1, 2, 3, 7):
        assert np.array_equal(np.asarray(s[i]), np.asarray(ref[i])), i
    assert np.array_equal(got["logits"], ref_logits)


def test_sharded_codec_at_bench_shape_equals_single_process(tmp_path):
    """What `bench.py --gpus N` runs per rank, checked on bytes: two ranks (gloo, sharing this box's GPU) with >= 100
    cubes of 64^3 each — every block takes the two-pipeline branch of compress_block / decompress_block with the early-z
    hook on a pipeline thread, on the 64^3 row kernels — against single-process compress_hyper on the whole 205-cube cloud:
    identical y strings, z string, ranges, point counts and occupancy masks."""
    from pcgcv1_amd import process, synthetic, transform
    from pcgcv1_amd.dataprocess import inout_points as iop
    from pcgcv1_amd.models import model_voxception as model
    cubes, _, nums = process.preprocess_points(synthetic.make_cloud(seed=1300), 1.0, 64, 64)
    B = int(cubes.shape[0])
    assert B >= 200
    ref = transform.compress_hyper(cubes, model, "synthetic:1300:sparse")
    assert transform.get_codec(model, "synthetic:1300:sparse").last_path["pipelines"] == 2
    ref_masks = iop.select_voxels(transform.decompress_hyper(*ref, model, "synthetic:1300:sparse"), nums, 1.0).cpu().numpy()
    ref_packed = np.packbits(ref_masks.reshape(B, -1), axis=1)
    got = _run(2, tmp_path, "gloo", n_cubes=0)
    s = got["stream"]
    assert got["path"] == {"call": "compress_block", "cubes": B - B // 2, "pipelines": 2}
    assert got["collectives"] == ["all_reduce z range", "gather z-hat", "all_reduce y bytes", "gather per-cube records", "gather y strings"]
    assert list(s[0]) == list(ref[0]) and s[4] == ref[4]
    for i in (1, 2, 3, 7):
        assert np.array_equal(np.asarray(s[i]), np.asarray(ref[i])), i
    assert (s[5], s[6]) == (ref[5], ref[6]) and np.array_equal(s[8], nums)
    assert np.array_equal(got["masks_packed"], ref_packed)


def test_every_collective_runs_through_rccl_on_one_rank_in_the_two_rank_order(tmp_path):
    """RCCL readiness without a second GPU: the bench-shaped sharded round trip (205 cubes of 64^3: two host pipelines, the
    z leg issued from a pipeline thread with PCGC_EARLY_Z=1) on a ONE-rank RCCL group — every collective of Exchange
    (all_reduce, gather, broadcast, all_gather_into_tensor) runs through RCCL on HBM buffers — gives the bytes of the
    two-rank gloo run, and its collective log (names, in order) is the two-rank one.  N > 1 over RCCL has never run:
    this box has one GPU (multi-GPU)."""
    two = _run(2, tmp_path, "gloo", n_cubes=0)
    one = _run(1, tmp_path, "nccl", n_cubes=0)
    assert one["collective_device"] == "cuda" and two["collective_device"] == "cpu" and one["early_z"] and two["early_z"]
    enc = ["all_reduce z range", "gather z-hat", "all_reduce y bytes", "gather per-cube records", "gather y strings"]
    dec = ["broadcast header", "broadcast z string", "broadcast per-cube records", "broadcast y strings", "all_reduce decode status"]
    for got in (one, two):
        assert got["collectives"] == enc
        assert got["collectives_decode"] == dec + ["gather occupancy bit masks"]
        assert got["collectives_gather_all"] == dec + ["all_gather occupancy bit masks"]
        assert all(b > 0 for b in got["bytes"])
    assert list(one["stream"][0]) == list(two["stream"][0]) and one["stream"][4] == two["stream"][4]
    assert np.array_equal(one["masks_packed"], two["masks_packed"])


def test_data_parallel_step(tmp_path):
    """train_hyper data parallelism: two ranks, each with its own batch, one all_reduce of the
    flat gradient buffer per step == the mean of the two replica gradients applied by one process."""
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / ("t%d.pkl" % world))
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_train_worker_gpu.py"), str(r), str(world), str(port), out,
                                   "gloo"]) for r in range(world)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        with open(out, "rb") as f:
            outs[world] = pickle.load(f)["weights"]
    for k in outs[1]:
        assert np.array_equal(outs[1][k], outs[2][k]), k


def test_cli_under_torchrun_writes_the_same_files(tmp_path):
    """test.py compress / decompress as two ranks (one GPU here, gloo) — under torch.distributed.run and through the CLI's own
    --gpu=2 — against the single-process run: identical container files and identical reconstructed ply."""
    from pcgcv1_amd import synthetic
    from pcgcv1_amd.dataprocess import inout_points as iop
    root = os.path.dirname(HERE)
    pts = synthetic.make_cloud(seed=15, res=256, n_shells=4, rmin=0.15, rmax=0.35)
    outs = {}
    for world in (1, 2, "--gpu=2"):                        # the last one: no launcher, the CLI starts its two ranks itself
        d = tmp_path / ("w%s" % str(world).strip("-"))
        d.mkdir()
        iop.write_ply_data(str(d / "c_vox8.ply"), pts)
        env = dict(os.environ, PCGC_BACKEND="gloo", PYTHONPATH=root)
        env.pop("WORLD_SIZE", None)
        launch = [sys.executable] if world != 2 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                                     "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                                                     "--master-port", str(_free_port())]
        common = ["--ckpt_dir=synthetic:7:sparse", "--min_num=20"] + ([world] if isinstance(world, str) else [])
        for cmd in (["compress", "c_vox8.ply"], ["decompress", "compressed/c_vox8"]):
            r = subprocess.run(launch + ["-m", "pcgcv1_amd.test"] + cmd + common, cwd=str(d), env=env, capture_output=True, timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs[world] = {f: (d / "compressed" / f).read_bytes() for f in sorted(os.listdir(d / "compressed"))}
        outs[world]["rec"] = (d / "c_vox8_rec.ply").read_bytes()
    for other in (2, "--gpu=2"):
        assert sorted(outs[1]) == sorted(outs[other])
        for k in outs[1]:
            assert outs[1][k] == outs[other][k], (other, k)


def test_bench_two_ranks_prints_one_json_line():
    """bench.py --gpus 2 exactly as the driver launches it (torchrun, default flags apart from a short run): both ranks
    exit 0 and rank 0 prints the contract's JSON line with the sharded-path extras.  The ranks share this box's one
    GPU over gloo (RCCL refuses two ranks on one device); the rank-0-only roofline block after the timed region must not
    enter a collective."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, PCGC_BENCH_BACKEND="gloo", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "cubes/s"
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d          # the CPU baseline is an N = 1 figure
    assert any("gather" in c["name"] for c in d["collectives"]) and d["strong_scaling"]["value"] > 0
    assert d["independent_clouds"]["value"] > 0
    assert d["ranks"] == {"world_size": 2, "backend": "gloo", "devices_visible": 1}


def test_bare_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (the form of the driver's N = 1 command) starts its two
    ranks itself instead of failing an assert, relays rank 0's JSON line and exits 0."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PCGC_BENCH_BACKEND="gloo", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-roofline"], cwd=root, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["value"] > 0


def test_sharded_codec_on_eight_ranks(tmp_path):
    """The driver's largest world: EIGHT ranks (gloo, sharing this box's GPU) on the 205-cube bench cloud at 64^3 — blocks of
    25 / 26 cubes, the geometric decoder blocks down to a handful — and on FIVE cubes (fewer cubes than ranks: three ranks'
    encoder blocks and most decoder blocks are empty) reproduce the single-process bytes, ranges, point counts and masks."""
    from pcgcv1_amd import process, synthetic, transform
    from pcgcv1_amd.dataprocess import inout_points as iop
    from pcgcv1_amd.models import model_voxception as model
    cubes, _, nums = process.preprocess_points(synthetic.make_cloud(seed=1300), 1.0, 64, 64)
    B = int(cubes.shape[0])
    ref = transform.compress_hyper(cubes, model, "synthetic:1300:sparse")
    ref_masks = iop.select_voxels(transform.decompress_hyper(*ref, model, "synthetic:1300:sparse"), nums, 1.0).cpu().numpy()
    got = _run(8, tmp_path, "gloo", n_cubes=0)
    s = got["stream"]
    assert got["collectives"] == ["all_reduce z range", "gather z-hat", "all_reduce y bytes", "gather per-cube records", "gather y strings"]
    assert list(s[0]) == list(ref[0]) and s[4] == ref[4]
    for i in (1, 2, 3, 7):
        assert np.array_equal(np.asarray(s[i]), np.asarray(ref[i])), i
    assert (s[5], s[6]) == (ref[5], ref[6]) and np.array_equal(s[8], nums)
    assert np.array_equal(got["masks_packed"], np.packbits(ref_masks.reshape(B, -1), axis=1))
    # five cubes on eight ranks
    cubes5 = synthetic.make_cubes(seed=9, n_cubes=5, cube_size=32, occupancy=0.03)
    ref5 = transform.compress_hyper(cubes5, model, "synthetic:21:dense")
    ref5_logits = transform.decompress_hyper(*ref5, model, "synthetic:21:dense").cpu().numpy()
    got5 = _run(8, tmp_path, "gloo", n_cubes=5)
    s5 = got5["stream"]
    assert list(s5[0]) == list(ref5[0]) and s5[4] == ref5[4]
    for i in (1, 2, 3, 7):
        assert np.array_equal(np.asarray(s5[i]), np.asarray(ref5[i])), i
    assert np.array_equal(got5["logits"], ref5_logits)


def test_bench_eight_ranks_dry_run():
    """The driver's SCALE command at N = 8, as far as one GPU can rehearse it: `python bench.py --gpus 8` bare (its own
    --standalone launcher; children only — nothing re-execs a process that touched the GPU) over gloo on the one device.
    Every rank exits 0, rank 0 prints ONE JSON line with world_size 8 and the collectives of a step, and the strong-scaling
    leg (the one 205-cube cloud cut into eight blocks) codes the bytes the single-process codec codes (CRC-32 of the y strings
    + z string, printed by the N = 1 line as config.stream_crc32).  NOT a scaling number: eight processes share one GPU."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PCGC_BENCH_BACKEND="gloo", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--cpu-cubes", "0"], cwd=root, env=env, capture_output=True, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks"] == {"world_size": 8, "backend": "gloo", "devices_visible": 1} and d["value"] > 0
    assert [c["name"] for c in d["collectives"]][:2] == ["all_reduce z range", "gather z-hat"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-extras", "--cpu-cubes", "0",
                         "--no-roofline"], cwd=root, env=env, capture_output=True, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    d1 = json.loads([l for l in r1.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["strong_scaling"]["stream_crc32"] == d1["config"]["stream_crc32"]
    out = os.environ.get("PCGC_SAVE_BENCH8")
    if out:
        d["note"] = "eight ranks over gloo sharing ONE GPU: a rehearsal of the N = 8 code path, not a scaling number"
        with open(out, "w") as f:
            f.write(json.dumps(d) + "\n")
