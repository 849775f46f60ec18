"""the N>1 path on CPU — world_size 2 (gloo, 127.0.0.1) must produce exactly the single-process
bitstream and reconstruction: contiguous cube blocks per rank, all_reduce of the z range, gathers to rank 0."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np

from pcgcv1_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 5, 8, 202, 2003):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp_path):
    out = str(tmp_path / ("w%d.pkl" % world))
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), str(r), str(world), str(port), out],
                              env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    with open(out, "rb") as f:
        return pickle.load(f)


def test_world2_equals_world1(tmp_path):
    one = _run(1, tmp_path)
    two = _run(2, tmp_path)
    s1, s2 = one["stream"], two["stream"]
    assert s1[0] == s2[0] and s1[4] == s2[4]                      # y strings (per cube, in order) and the single z string
    for i in (1, 2, 3, 7):
        assert np.array_equal(s1[i], s2[i])
    assert (s1[5], s1[6]) == (s2[5], s2[6])
    for run in (one, two):                                        # block-local form == whole-list form, + point counts
        sl = run["stream_local"]
        assert sl[0] == s1[0] and sl[4] == s1[4] and all(np.array_equal(sl[i], s1[i]) for i in (1, 2, 3, 7))
        assert sl[8].dtype == np.uint16 and sl[8].shape == (5,)
    assert np.array_equal(one["stream_local"][8], two["stream_local"][8])
    # world 2 moves everything through tensor collectives on pre-sized buffers (no pickled objects)
    names = [c[0] for c in two["collectives"]]
    assert names == ["all_reduce z range", "gather z-hat", "all_reduce y bytes", "gather per-cube records", "gather y strings"]
    assert all(c[1] > 0 and c[2] is not None for c in two["collectives"]) and one["collectives"] == []
    assert np.array_equal(one["logits"], two["logits"])
    assert np.array_equal(one["masks"], two["masks"]) and one["masks"].shape == (5, 16, 16, 16, 1)


def test_out_of_range_hyper_latents_raise_on_every_rank(tmp_path):
    """A rank whose z-hat leaves int8 must not abandon its peers inside the all_reduce: the range is validated after the
    collective, on the global min / max, so both ranks raise OverflowError (exit code 7 of the worker) and none hangs."""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), str(r), "2", str(port),
                               str(tmp_path / "x.pkl"), "overflow"]) for r in range(2)]
    assert [p.wait(timeout=300) for p in procs] == [7, 7]


def test_a_local_failure_reaches_every_rank_at_the_next_collective(tmp_path):
    """One rank fails in its LOCAL part — before the z leg, after the z leg (its y strings; also with an OverflowError, the
    type the z leg's own collective verdict has), between the z all_reduce and the z-hat gather, in its decoder block: it takes
    part in the next collective with a poisoned status word and re-raises (worker exit code 8); its peer raises
    sharding.PeerFailure right after that collective (exit code 9).  Neither waits for a timeout."""
    import time
    for mode in ("fail_before_z", "fail_after_z", "fail_after_z_overflow", "fail_gather_buffer", "fail_decode"):
        port = _free_port()
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), str(r), "2", str(port),
                                   str(tmp_path / "x.pkl"), mode]) for r in range(2)]
        assert [p.wait(timeout=300) for p in procs] == [9, 8], mode
        assert time.time() - t0 < 120, mode


def test_decode_ranges_cover_the_cubes_in_order():
    """sharding.decode_ranges: contiguous, order-preserving, sizes sum to n and never grow with the rank (a later rank
    waits longer for its z symbols); rho = 1 reproduces shard_range."""
    from pcgcv1_amd import sharding
    for n in (0, 1, 3, 5, 26, 205, 1640):
        for world in (1, 2, 3, 8):
            r = sharding.decode_ranges(n, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
            sizes = [h - l for l, h in r]
            assert all(s >= 0 for s in sizes) and all(a >= b for a, b in zip(sizes[:-1], sizes[1:]))
            assert sharding.decode_ranges(n, world, rho=1.0) == [sharding.shard_range(n, k, world) for k in range(world)]
    sizes = [h - l for l, h in sharding.decode_ranges(1640, 8, rho=0.894)]
    assert sizes[0] > 2 * sizes[-1] and sum(sizes) == 1640
