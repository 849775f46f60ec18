import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcgcv1_amd import synthetic, checkpoint, transform
from pcgcv1_amd.models import model_voxception as model
w = synthetic.make_weights(seed=3, profile="sparse")
checkpoint._CACHE["t"] = w
c = transform.get_codec(model, "t")
x = torch.from_numpy(synthetic.make_cubes(seed=3, n_cubes=int(sys.argv[1]) if len(sys.argv) > 1 else 2, cube_size=64)).cuda()
y = c.analysis_transform(x)
torch.cuda.synchronize()
print("ok", float(y.abs().sum()))
np.save("/tmp/y_%s.npy" % os.environ.get("PCGC_ROW_STAGES", "31"), y.cpu().numpy())
