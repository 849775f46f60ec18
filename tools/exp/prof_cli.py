import cProfile, pstats, os, sys, io, contextlib, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pcgcv1_amd import checkpoint, synthetic, test as cli
from pcgcv1_amd.dataprocess import inout_points as iop
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
d = tempfile.mkdtemp(); os.chdir(d)
iop.write_ply_data("cloud_vox10.ply", pts)
for _ in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(["compress", "cloud_vox10.ply", "--ckpt_dir=bench"]); cli.main(["decompress", "compressed/cloud_vox10", "--ckpt_dir=bench"])
import torch
for rep in range(6):
    ts = []
    for cmd in (["compress", "cloud_vox10.ply", "--ckpt_dir=bench"], ["decompress", "compressed/cloud_vox10", "--ckpt_dir=bench"]):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(cmd)
        ts.append(1e3 * (time.perf_counter() - t0))
    print("compress %.1f ms  decompress %.1f ms  -> %.0f cubes/s" % (ts[0], ts[1], 205e3 / sum(ts)))
for cmd in (["compress", "cloud_vox10.ply", "--ckpt_dir=bench"], ["decompress", "compressed/cloud_vox10", "--ckpt_dir=bench"]):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable(); cli.main(cmd); torch.cuda.synchronize(); pr.disable()
    print(cmd[0], "%.1f ms" % (1e3 * (time.perf_counter() - t0)))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print("\n".join(s.getvalue().splitlines()[6:56]))
