"""metrics.pc_error_off_grid (the eval harness' D1 / D2 for a rate section with scale != 1, whose reconstruction has
fractional coordinates) against known answers of the prebuilt pc_error_d (tools/make_offgrid_golden.py, build container)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "pc_error_offgrid.npz")


@pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/pc_error_offgrid.npz not present")
def test_off_grid_pc_error_matches_the_prebuilt_binary():
    from pcgcv1_amd import metrics
    g = np.load(GOLD)
    keys = [str(k) for k in g["keys"]]
    for i in range(int(g["n_cases"])):
        a, na, b, res = g["a%d" % i], g["na%d" % i], g["b%d" % i], int(g["res%d" % i])
        assert metrics._off_grid(b) and not metrics._off_grid(a.astype(np.float32))
        mine = metrics.pc_error(a, b, na, res - 1)                 # dispatches on the fractional coordinates
        for k, want in zip(keys, g["vals%d" % i]):
            tol = 1e-3 if "PSNR" in k else 2e-5 * max(1.0, abs(float(want)))      # pc_error prints 6 significant digits
            assert abs(mine[k] - float(want)) < tol, (i, k, mine[k], float(want))
    # an integral "float" cloud is NOT off grid (scale 0.5 scales back to integers): the device path takes it
    assert not metrics._off_grid(np.array([[1.0, 2.0, 4.0]], np.float32))
