"""Per-launch times of the analysis' 64^3 stage with empty-space skipping against the number of tiles each launch computes
(host restatement of tile_order_kernel's rule), for several chunk sizes: where do skipped launches lose time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
cubes = cubes[:int(os.environ.get('NB', '205'))].contiguous()
occ = (cubes.reshape(cubes.shape[0], 64, 64, 64) != 0).any(dim=3).cpu().numpy()
c = np.zeros((occ.shape[0], 65, 65), np.int64); c[:, 1:, 1:] = occ.cumsum(1).cumsum(2)
def heavy(lo_c, hi_c, r):
    n = 0
    for d0 in range(0, 64, 8):
        dl, dh = max(d0 - r, 0), min(d0 + 7 + r, 63)
        for h0 in range(0, 64, 2):
            hl, hh = max(h0 - r, 0), min(h0 + 1 + r, 63)
            n += int(((c[lo_c:hi_c, dh + 1, hh + 1] - c[lo_c:hi_c, dl, hh + 1] - c[lo_c:hi_c, dh + 1, hl] + c[lo_c:hi_c, dl, hl]) > 0).sum())
    return n
net = transform.get_codec(model, "bench").analysis_transform
def run(env, show):
    for k, v in env.items(): os.environ[k] = v
    for _ in range(2): net(cubes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): net(cubes)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 5
    net.set_profiling(True); net(cubes); rows = net.profile_report(); net.set_profiling(False)
    r64 = [r for r in rows if r["Din"] == 64 and r["kernel"] in ("rowA", "rowBC")]
    print("%s: analysis of %d cubes %.2f ms; 64^3 A+BC launches %d, %.2f ms (profiled, serial)" % (env, len(cubes), tot, len(r64), sum(r["ms"] for r in r64)))
    if show:
        at, i = 0, 0
        while i < len(r64):
            B = r64[i]["B"]
            line = "  cubes %3d..%3d:" % (at, at + B)
            for j in range(6):
                r = r64[i + j]
                H = heavy(at, at + B, 2 + j)
                line += "  %s %5.1f us H=%4d (%.2f)" % ("A " if j % 2 == 0 else "BC", 1e3 * r["ms"], H, H / 2048.0)
            print(line)
            at += B; i += 6
chs = sys.argv[1].split(":") if len(sys.argv) > 1 else ("8", "12", "16", "20", "24", "32")
for ch in chs:
    run({"PCGC_CHUNKS_A": ch + ",64,256"}, ch in ("16", "32"))
run({"PCGC_CHUNKS_A": "8,64,256", "PCGC_SKIP_EMPTY": "0"}, False)
