"""oracle/points.py against the golden vectors produced by the reference's own numpy code
(tools/make_golden.py -> tests/golden/partition.npz, select.npz)."""
import numpy as np
import pytest

from oracle import points as op

CASES = ["a", "b", "c", "d"]


def _split(flat, lens):
    return np.split(flat, np.cumsum(lens)[:-1])


@pytest.mark.parametrize("name", CASES)
def test_preprocess_matches_reference(golden, name):
    g = golden("partition.npz")
    cube, min_num, scale = g[name + "_args"]
    cubes, pos, nums = op.preprocess_points(g[name + "_points"], float(scale), int(cube), int(min_num))
    assert np.array_equal(pos, g[name + "_cube_positions"])            # first-appearance order
    assert nums.dtype == np.uint16 and np.array_equal(nums, g[name + "_points_numbers"])
    occ = _split(g[name + "_occ_flat"], g[name + "_occ_lens"])
    assert len(occ) == len(cubes)
    for c, o in zip(cubes, occ):
        assert np.array_equal(np.flatnonzero(c), o)                     # sorted-key cube order + voxel indices


@pytest.mark.parametrize("name", CASES)
def test_ply_writer_and_identity_roundtrip(golden, name, tmp_path):
    g = golden("partition.npz")
    cube, min_num, scale = g[name + "_args"]
    assert op.ply_text(g[name + "_points"]).encode() == g[name + "_ply"].tobytes()
    f = tmp_path / "x.ply"
    f.write_bytes(g[name + "_ply"].tobytes())
    assert np.array_equal(op.load_ply_data(str(f)), g[name + "_points"])
    cubes, pos, nums = op.preprocess_points(g[name + "_points"], float(scale), int(cube), int(min_num))
    rec = op.postprocess_points(cubes, nums, pos, float(scale), int(cube), 1.0)
    assert op.ply_text(rec).encode() == g[name + "_rec_ply"].tobytes()


def test_select_voxels_and_merge(golden):
    g = golden("select.npz")
    vols, nums = g["vols"], g["nums"]
    for rho in (1.0, 1.1, 0.5):
        assert np.array_equal(op.select_voxels(vols, nums, rho).astype(np.uint8), g["mask_rho%g" % rho])
    assert np.array_equal(op.select_voxels(vols, nums, 1.0, fixed_thres=0.0).astype(np.uint8), g["mask_fixed0"])
    pts = op.voxels2points(g["mask_rho1"])
    assert [len(p) for p in pts] == list(g["v2p_lens"])
    assert np.array_equal(np.concatenate(pts), g["v2p_flat"])
    merged = op.merge_points(pts, g["merge_positions"], 16)
    assert op.ply_text(merged).encode() == g["merge_ply"].tobytes()
    assert op.ply_text(g["float_points"]).encode() == g["float_ply"].tobytes()


def test_edge_cases():
    with pytest.raises(ValueError):
        op.partition(np.array([[1, 2, 3]] * 5), 64, 20)                 # nothing survives min_num
    pts = np.array([[0, 0, 0]] * 3 + [[65, 1, 1]] * 2)
    sp, pos = op.partition(pts, 64, 2)
    assert pos.tolist() == [[0, 0, 0], [1, 0, 0]] and [len(s) for s in sp] == [3, 2]
    vox = op.points2voxels(sp, 64)                                      # duplicate points collapse
    assert vox.sum() == 2
    assert len(op.voxels2points(vox[:1])) == 1                          # B == 1 works here
