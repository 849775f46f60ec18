import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pcgcv1_amd import synthetic, process, metrics, eval as pe
from pcgcv1_amd.dataprocess import inout_points as iop
from pcgcv1_amd.models import model_voxception as model
t=time.time(); pts = synthetic.make_cloud(seed=2000); print("make_cloud", time.time()-t)
c = pts.mean(0); nrm = (pts - c) / np.maximum(np.linalg.norm(pts - c, axis=1, keepdims=True), 1e-9)
t=time.time()
with open("/tmp/f.ply","w") as fh:
    fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\nend_header\n" % len(pts))
    np.savetxt(fh, np.concatenate([pts.astype(np.float64), nrm], 1), fmt="%d %d %d %.6f %.6f %.6f")
print("write ply", time.time()-t)
t=time.time(); p2, n2 = iop.load_ply_normals("/tmp/f.ply"); print("load ply normals", time.time()-t)
for rep in range(2):
    t=time.time(); cubes_d, pos, nums, n, bpps = pe.rate_point(pts, model, "synthetic:41:sparse", 1.0, 64, 64); torch.cuda.synchronize(); print("rate_point", time.time()-t)
t=time.time(); rec = process.postprocess_points(cubes_d, nums, pos, 1.0, 64, 1.0); rec = np.unique(np.rint(rec).astype(np.int32), axis=0); print("postprocess+unique", time.time()-t)
for rep in range(2):
    t=time.time(); r = metrics.pc_error(pts, rec, nrm, 1023); print("pc_error", time.time()-t)
t=time.time(); r = metrics.d1_metrics(pts, rec, 1023); print("d1 only", time.time()-t)
