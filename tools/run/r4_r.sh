#!/bin/bash
mkdir -p gpurun_out/r4r
python -m pytest tests/test_trained_checkpoints.py -m gpu -q -x -s -k golden > gpurun_out/r4r/gold.txt 2>&1; grep -a "full cloud\|passed\|failed" gpurun_out/r4r/gold.txt | cut -c1-400
