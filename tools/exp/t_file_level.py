"""In-process stage timing of the CLI path (warm): ply parse, partition + voxelise, codec, container write / read, codec, top-k, points, ply write."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.dataprocess import inout_points as iop, inout_bitstream as bs
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
d = tempfile.mkdtemp(); os.chdir(d)
pts = synthetic.make_cloud(seed=1300)
iop.write_ply_data("c.ply", pts)
def T(f, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)
for rep in range(4):
    rows = []
    p, t = T(iop.load_ply_data, "c.ply"); rows.append(("load_ply", t))
    (cubes, pos, nums), t = T(process.preprocess_points, p, 1.0, 64, 64); rows.append(("partition+voxelise", t))
    out, t = T(transform.compress_hyper, cubes, model, "bench"); rows.append(("compress_hyper", t))
    _, t = T(bs.write_binary_files_hyper, "c", out[0], out[4], nums, pos, out[1], out[2], out[3], out[5], out[6], out[7], rootdir="./compressed", verbose=False); rows.append(("write files", t))
    r, t = T(bs.read_binary_files_hyper, "c", "./compressed"); rows.append(("read files", t))
    xs, t = T(transform.decompress_hyper, r[0], r[4], r[5], r[6], r[1], r[7], r[8], r[9], model, "bench"); rows.append(("decompress_hyper", t))
    mask, t = T(iop.select_voxels, xs, r[2], 1.0); rows.append(("top-k", t))
    rec, t = T(iop.voxels2merged_points, mask, r[3], 64); rows.append(("voxels->points", t))
    _, t = T(iop.write_ply_data, "rec.ply", rec); rows.append(("write ply", t))
    if rep == 3:
        for n, t in rows: print("%-20s %7.2f ms" % (n, t))
        print("total %.1f ms" % sum(t for _, t in rows))
