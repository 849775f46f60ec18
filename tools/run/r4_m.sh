#!/bin/bash
mkdir -p gpurun_out/r4m
python tools/exp/t_file_level.py > gpurun_out/r4m/file_level.txt 2>&1; tail -11 gpurun_out/r4m/file_level.txt
python - > gpurun_out/r4m/bench_file.txt 2>&1 <<'P'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench, torch
from pcgcv1_amd import checkpoint, synthetic, process
checkpoint._CACHE["bench"] = checkpoint.load("checkpoints/hyper/a6.00b3.00")
pts = synthetic.make_cloud(seed=1300)
for _ in range(3):
    print(json.dumps(bench._file_level(pts, 205)))
P
tail -3 gpurun_out/r4m/bench_file.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "preprocess or cli or file or ply" > gpurun_out/r4m/t.txt 2>&1; tail -2 gpurun_out/r4m/t.txt
