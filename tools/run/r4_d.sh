#!/bin/bash
# round 4, call D: virtual empty tiles (PCGC_SKIP_EMPTY=1) against copy mode (2) and off (0)
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or slot_invariant or config3" -s > gpurun_out/r4d/t_skip.txt 2>&1
tail -4 gpurun_out/r4d/t_skip.txt; grep -a "frame [0-9]:\|config 3\|empty-space" gpurun_out/r4d/t_skip.txt
bash tools/run/ab_env.sh gpurun_out/r4d 3 "PCGC_SKIP_EMPTY=0" "PCGC_SKIP_EMPTY=2" "PCGC_SKIP_EMPTY=1" "PCGC_SKIP_EMPTY=1 PCGC_CHUNKS_A=24,64,256" > gpurun_out/r4d/ab.txt 2>&1
cat gpurun_out/r4d/ab.txt
PCGC_SKIP_EMPTY=1 python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4d/bench_skip1.json 2>gpurun_out/r4d/bench_skip1.err
