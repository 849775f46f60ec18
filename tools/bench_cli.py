"""Wall-clock of the whole file-level path (test.py compress / decompress, reference stage names) on the
synthetic longdress-like cloud, incl. ply parse / write and the container.  GPU box only."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcgcv1_amd import synthetic, test as cli
from pcgcv1_amd.dataprocess import inout_points as iop

d = tempfile.mkdtemp(prefix="pcgc_cli_")
os.chdir(d)
pts = synthetic.make_cloud(seed=1300)
t = time.time(); iop.write_ply_data("cloud_vox10.ply", pts); print("write input ply (%d points): %.2f s" % (len(pts), time.time() - t))
for rep in range(2):
    t = time.time(); cli.main(["compress", "cloud_vox10.ply", "--ckpt_dir=synthetic:1300:sparse"]); tc = time.time() - t
    t = time.time(); cli.main(["decompress", "compressed/cloud_vox10", "--ckpt_dir=synthetic:1300:sparse"]); td = time.time() - t
    print("run %d: compress %.2f s, decompress %.2f s" % (rep, tc, td))
size = sum(os.path.getsize(os.path.join("compressed", f)) for f in os.listdir("compressed"))
rec = iop.load_ply_data("cloud_vox10_rec.ply")
print("bytes %d -> bpp %.4f ; reconstructed points %d" % (size, 8.0 * size / len(pts), len(rec)))
