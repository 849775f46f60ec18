"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

Loop-level numpy restatement of the reference's integer point/cube handling:

  dataprocess/inout_points.py:8-28     load_ply_data
  dataprocess/inout_points.py:30-46    write_ply_data
  dataprocess/inout_points.py:50-90    load_points   (partition)
  dataprocess/inout_points.py:92-112   save_points   (merge)
  dataprocess/inout_points.py:116-132  points2voxels
  dataprocess/inout_points.py:134-143  voxels2points
  dataprocess/inout_points.py:147-179  select_voxels / get_adaptive_thres
  process.py:16-52                     preprocess
  process.py:54-82                     postprocess

PINNED: these reference modules are pure numpy and import in the build
container; tools/make_golden.py runs them on seeded inputs and commits the
results under tests/golden/ (partition_*.npz, select_*.npz, ply_*.txt).
tests/test_oracle_points.py checks this restatement against those vectors
bit-exactly.  Written as per-point / per-cube loops on purpose (small cases
only): the product's vectorised implementation is checked against it.

Deliberate non-reproductions of reference crashes (documented, not relied on):
a cube holding exactly one point is a 1-D array in the reference and counts as
"3 points" (inout_points.py:66, 72); here it counts as 1 point.  A batch of
exactly one cube breaks np.squeeze in voxels2points (inout_points.py:137);
here B=1 works.
"""
import numpy as np


def load_ply_data(filename):
    pts = []
    with open(filename) as f:
        for line in f:
            w = line.split(" ")
            try:
                pts.append([float(w[0]), float(w[1]), float(w[2])])
            except (ValueError, IndexError):
                continue
    return np.array(pts).reshape(-1, 3).astype(np.int32)


def ply_text(points):
    """The exact bytes write_ply_data (inout_points.py:30-46) produces."""
    points = np.asarray(points)
    lines = ["ply\n", "format ascii 1.0\n", "element vertex %d\n" % points.shape[0],
             "property float x\n", "property float y\n", "property float z\n", "end_header\n"]
    for p in points:
        lines.append("%s %s %s\n" % (str(p[0]), str(p[1]), str(p[2])))
    return "".join(lines)


def write_ply_data(filename, points):
    with open(filename, "w") as f:
        f.write(ply_text(points))


def order_key(cube_positions):
    """inout_points.py:80-82 / 96-98."""
    cube_positions = np.asarray(cube_positions)
    step = cube_positions.max() + 1
    return cube_positions[:, 0] + cube_positions[:, 1] * step + cube_positions[:, 2] * step * step, step


def ordered_positions(cube_positions):
    key, step = order_key(cube_positions)
    key = np.sort(key)
    return np.stack([key % step, (key // step) % step, key // step // step], -1)


def partition(point_cloud, cube_size=64, min_num=20):
    """load_points minus the file read. Returns (set_points in key order [int16],
    cube_positions in first-appearance order [int64, n x 3])."""
    cubes = {}
    for p in np.asarray(point_cloud):
        idx = tuple(int(v) for v in (p // cube_size))
        cubes.setdefault(idx, []).append(p % cube_size)
    cubes = {k: np.array(v) for k, v in cubes.items() if len(v) >= min_num}
    if not cubes:
        raise ValueError("no cube holds at least min_num points")      # reference: .max() of empty array
    cube_positions = np.array(list(cubes.keys()))
    set_points = [cubes[tuple(int(v) for v in k)].astype(np.int16) for k in ordered_positions(cube_positions)]
    return set_points, cube_positions


def points2voxels(set_points, cube_size):
    vox = np.zeros((len(set_points), cube_size, cube_size, cube_size, 1))
    for i, pts in enumerate(set_points):
        pts = pts.astype(int)
        vox[i, pts[:, 0], pts[:, 1], pts[:, 2], 0] = 1.0
    return vox


def voxels2points(voxels):
    voxels = np.uint8(np.asarray(voxels))
    voxels = voxels.reshape(voxels.shape[:4])
    return [np.array(np.where(v > 0)).transpose((1, 0)) for v in voxels]


def adaptive_threshold(vol, num, init_thres=-2.0):
    values = vol[vol > init_thres]
    if values.shape[0] < num:
        values = np.reshape(vol, [-1])
    values = np.sort(values)
    return values[-num]          # num == 0 -> values[0] (inout_points.py:177)


def select_voxels(vols, points_nums, offset_ratio=1.0, fixed_thres=None):
    masks = []
    for i, vol in enumerate(vols):
        if fixed_thres is None:
            thres = adaptive_threshold(vol, int(offset_ratio * np.array(points_nums[i])))
        else:
            thres = fixed_thres
        masks.append(np.greater_equal(vol, thres).astype("float32"))
    return np.stack(masks)


def merge_points(set_points, cube_positions, cube_size=64):
    """save_points minus the file write."""
    out = [v + np.array(k) * cube_size for k, v in zip(ordered_positions(cube_positions), set_points)]
    return np.concatenate(out).astype("int")


def preprocess_points(points, scale, cube_size, min_num):
    """process.py:16-52 on an in-memory cloud. Returns (cubes f64, cube_positions, points_numbers u16)."""
    if scale != 1:
        down = np.round(points.astype("float32") * scale)
        points = np.unique(down, axis=0).astype(np.int32)     # ply text round trip keeps integers
    set_points, cube_positions = partition(points, cube_size, min_num)
    cubes = points2voxels(set_points, cube_size)
    points_numbers = np.sum(cubes, axis=(1, 2, 3, 4)).astype(np.uint16)
    return cubes, cube_positions, points_numbers


def postprocess_points(cubes, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres=None):
    """process.py:54-82 up to (not including) the ply write. Returns the point array written."""
    mask = select_voxels(cubes, points_numbers, rho, fixed_thres=fixed_thres)
    pts = merge_points(voxels2points(mask), cube_positions, cube_size)
    if scale == 1:
        return pts
    return pts.astype(np.int32).astype("float32") * float(1 / scale)
