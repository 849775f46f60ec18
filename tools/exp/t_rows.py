"""Per-kernel times of the two transforms on the bench cloud's cubes, single stream, one process — for A/B of kernel variants
selected by environment knobs that the launchers read per call:

    python tools/exp/t_rows.py [N_CUBES=103] [REPS=6] -- "" "PCGC_SKIP_EMPTY=0" ...
(round 5 used a temporary per-launch knob, PCGC_ROW_VARIANT, to flip kernel template variants this way: profiles/r05_vA_row_variants.txt)

For every setting (applied with os.environ inside this process, in interleaved rounds) the analysis and the synthesis run
REPS times with pcgc_net profiling on; prints mean us per launch and ms per forward for every (net, kernel, layer group).
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from pcgcv1_amd import checkpoint, process, synthetic, transform
from pcgcv1_amd.models import model_voxception as model

args = sys.argv[1:]
settings = [""]
if "--" in args:
    k = args.index("--")
    settings = args[k + 1:] or [""]
    args = args[:k]
n_cubes = int(args[0]) if len(args) > 0 else 103
reps = int(args[1]) if len(args) > 1 else 6
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
checkpoint._CACHE["bench"] = checkpoint.load(os.path.join(root, "checkpoints", "hyper", "a6.00b3.00"))
pts = synthetic.make_cloud(seed=1300)
cubes, pos, nums = process.preprocess_points(pts, 1.0, 64, 64)
x = (cubes[:n_cubes] if torch.is_tensor(cubes) else torch.from_numpy(np.ascontiguousarray(cubes[:n_cubes], np.float32))).cuda().float().contiguous()
c = transform.get_codec(model, "bench")
nets = {"analysis": c.analysis_transform, "synthesis": c.synthesis_transform}
y = c.analysis_transform(x)
yq = torch.round(y)
ref = {"analysis": y.clone(), "synthesis": c.synthesis_transform(yq).clone()}
torch.cuda.synchronize()


def apply(setting):
    for kv in setting.split():
        k_, v_ = kv.split("=", 1)
        os.environ[k_] = v_


def unapply(setting):
    for kv in setting.split():
        os.environ.pop(kv.split("=", 1)[0], None)


res = {s: collections.OrderedDict() for s in settings}
same = {s: True for s in settings}
for rnd in range(2):
    for s in settings:
        apply(s)
        for name, net in nets.items():
            inp = x if name == "analysis" else yq
            for _ in range(2):
                out = net(inp)
            same[s] = same[s] and bool(torch.equal(out, ref[name]))
            net.set_profiling(True)
            for _ in range(reps):
                net(inp)
            torch.cuda.synchronize()
            for r in net.profile_report():
                key = (name, r["kernel"], r["Din"], r["cin"], r["cout"], r["B"])
                a = res[s].setdefault(key, [0.0, 0])
                a[0] += r["ms"]
                a[1] += 1
            net.set_profiling(False)
        unapply(s)
print("%d cubes, %d forwards per setting and net" % (n_cubes, 2 * reps))
keys = list(res[settings[0]].keys())
print("%-58s" % "net kernel D cin cout B" + "".join("  %26s" % (s or "default")[-26:] for s in settings))
tot = {s: {"analysis": 0.0, "synthesis": 0.0} for s in settings}
for key in keys:
    line = "%-58s" % (" ".join(str(v) for v in key))
    for s in settings:
        ms, n = res[s].get(key, [0.0, 0])
        per_fwd = ms / (2 * reps)
        tot[s][key[0]] += per_fwd
        line += "  %9.1f us x%3d %7.3f ms" % (1e3 * ms / max(n, 1), n // (2 * reps), per_fwd)
    print(line)
for name in nets:
    print("%-58s" % (name + " total ms per forward") + "".join("  %26.3f" % tot[s][name] for s in settings))
print("outputs bit-identical to the default setting's first run:", same)
