#!/bin/bash
# A/B of the down_1 row-kernel launch geometries (PCGC_DOWN1) on one box
PCGC_ROW_STAGES=15 timeout 120 python tools/exp/t_down.py 2>&1 | tail -1
PCGC_ROW_STAGES=31 timeout 120 python tools/exp/t_down.py 2>&1 | tail -1
python -c "
import numpy as np
a=np.load('/tmp/y_15.npy'); b=np.load('/tmp/y_31.npy'); print('maxdiff', np.abs(a-b).max(), 'scale', np.abs(a).max())"
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "transforms or roundtrip or batch_slot" 2>&1 | tail -2
for v in 0 1 2 3 4; do
  PCGC_DOWN1=$v PCGC_BENCH_TOP=10 python bench.py --no-extras --cpu-cubes 0 > gpurun_out/dn_$v.json 2>/dev/null
  python tools/print_bench.py gpurun_out/dn_$v.json | grep "gpurun\|down_row\|mode=1>@D64"
done
