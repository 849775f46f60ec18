#!/bin/bash
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or config3" -s > gpurun_out/r4e/t.txt 2>&1
tail -4 gpurun_out/r4e/t.txt; grep -a "frame [0-9]:\|config 3\|empty-space" gpurun_out/r4e/t.txt
