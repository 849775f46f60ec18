"""Wall-clock timeline of the CLI decompress / compress (warm): when each stage starts and ends, per thread.
    python tools/exp/t_cli_timeline.py [min_us]"""
import contextlib, io, os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pcgcv1_amd import _lib, checkpoint, process, synthetic, transform, test as cli
from pcgcv1_amd.dataprocess import inout_points as iop, inout_bitstream as bs
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
LOG, T0, ON = [], [0.0], [False]


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        if not ON[0]:
            return f(*a, **k)
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, label))
    setattr(obj, name, g)


for mod, names in ((iop, ("load_ply_data", "partition", "voxelize_partition", "select_voxels", "voxels2merged_points", "_ply_parts", "ordered_positions")),
                   (process, ("preprocess", "preprocess_points", "_pwrite_all")), (process.StreamedPostprocess, ("_slice", "finish", "__init__")),
                   (transform, ("compress_hyper", "decompress_hyper")),
                   (bs, ("write_binary_files_hyper", "read_binary_files_hyper", "encode_cube_positions", "decode_cube_positions"))):
    for n in names:
        if hasattr(mod, n):
            wrap(mod, n)
_lib._trace = lambda label: ON[0] and LOG.append((time.perf_counter() - T0[0], time.perf_counter() - T0[0], threading.current_thread().name, "mark: " + label))
d = tempfile.mkdtemp(); os.chdir(d)
iop.write_ply_data("cloud_vox10.ply", synthetic.make_cloud(seed=1300))
cmds = (["compress", "cloud_vox10.ply", "--ckpt_dir=bench"], ["decompress", "compressed/cloud_vox10", "--ckpt_dir=bench"])
for _ in range(3):
    for c in cmds:
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(c)
for c in cmds:
    LOG.clear(); ON[0] = True; torch.cuda.synchronize(); T0[0] = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(c)
    total = time.perf_counter() - T0[0]; ON[0] = False
    print("==== %s: %.2f ms" % (c[0], 1e3 * total))
    for t0, t1, th, label in sorted(LOG):
        if (t1 - t0) * 1e6 >= min_us or label.startswith("mark"):
            print("%8.2f %8.2f %7.2f ms  %-14s %s" % (1e3 * t0, 1e3 * t1, 1e3 * (t1 - t0), th[-14:], label))
