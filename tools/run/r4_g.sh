#!/bin/bash
# round 4, call G: Q4 training layout; config 3 with the re-derived a6b3; the regenerated full-cloud golden
mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_train.py -q -x -s -k "q4 or full_size or config4 or autograd" > gpurun_out/r4g/t_train.txt 2>&1
tail -6 gpurun_out/r4g/t_train.txt
for q in 1 0 1 0; do PCGC_TRAIN_Q4=$q python tools/bench_train.py 20 2>&1 | tail -1 | sed "s/^/PCGC_TRAIN_Q4=$q /"; done | tee gpurun_out/r4g/bench_train.txt
python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/r4g/t_train_all.txt 2>&1; tail -3 gpurun_out/r4g/t_train_all.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "config3" -s > gpurun_out/r4g/t_c3.txt 2>&1
tail -3 gpurun_out/r4g/t_c3.txt; grep -a "frame [0-9]:\|config 3" gpurun_out/r4g/t_c3.txt
python -m pytest tests/test_trained_checkpoints.py -q -x -m gpu -s > gpurun_out/r4g/t_trained.txt 2>&1
tail -3 gpurun_out/r4g/t_trained.txt; grep -a "full cloud vs" gpurun_out/r4g/t_trained.txt
