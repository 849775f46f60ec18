#!/bin/bash
mkdir -p gpurun_out/r4l
PCGC_PART_TIMES=1 python tools/exp/t_file_level.py > gpurun_out/r4l/file_level.txt 2>&1; grep -a "partition " gpurun_out/r4l/file_level.txt | tail -6; tail -11 gpurun_out/r4l/file_level.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "preprocess or cli or file" > gpurun_out/r4l/t.txt 2>&1; tail -2 gpurun_out/r4l/t.txt
