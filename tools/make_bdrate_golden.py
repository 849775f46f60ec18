"""tests/golden/bdrate_cases.json: (curve1, curve2, bdsnr, bdrate) from the reference's myutils/bdrate_metrics.py, run in
the build container (numpy only; /root/reference does not exist on the GPU box, the vectors travel instead).

    python tools/make_bdrate_golden.py [/root/reference]
"""
import json
import os
import sys

import numpy as np


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    sys.dont_write_bytecode = True              # nothing is written next to the reference's sources
    sys.path.insert(0, ref_root)
    from myutils import bdrate_metrics as ref
    rng = np.random.default_rng(7)
    # this repository's six trained rate points (checkpoints/hyper/README.md) against a shifted copy and against a
    # part of itself, seeded monotone curves of 4-8 points, and two curves whose PSNR ranges do not meet
    base = [(0.0754, 69.58), (0.1113, 71.05), (0.1225, 71.64), (0.1512, 72.14), (0.1768, 72.34), (0.2044, 72.41)]
    cases = [(base, [(r * 1.3, p - 0.4) for r, p in base]), (base[:4], base[2:])]
    for n in (4, 5, 6, 8):
        for _ in range(3):
            r1, p1 = np.sort(rng.uniform(0.05, 1.5, n)), np.sort(rng.uniform(55, 78, n))
            r2, p2 = np.sort(rng.uniform(0.05, 1.5, n)), np.sort(rng.uniform(55, 78, n))
            cases.append((list(zip(r1.tolist(), p1.tolist())), list(zip(r2.tolist(), p2.tolist()))))
    cases.append(([(0.1, 60.0), (0.2, 65.0), (0.4, 70.0), (0.8, 72.0)], [(0.1, 10.0), (0.2, 12.0), (0.4, 13.0), (0.8, 14.0)]))
    out = [{"curve1": c1, "curve2": c2, "bdsnr": float(ref.bdsnr(c1, c2)), "bdrate": float(ref.bdrate(c1, c2))} for c1, c2 in cases]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bdrate_cases.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
