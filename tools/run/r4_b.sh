#!/bin/bash
# round 4, call B: heavy-first tile order for the empty-space skipping
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or slot_invariant" -s > gpurun_out/r4b/t_skip.txt 2>&1
python -m pytest tests/test_trained_checkpoints.py -q -x -m gpu -s -k full_cloud > gpurun_out/r4b/t_golden.txt 2>&1
bash tools/run/ab_env.sh gpurun_out/r4b 3 "PCGC_SKIP_EMPTY=0" "PCGC_SKIP_EMPTY=1" "PCGC_SKIP_EMPTY=1 PCGC_CHUNKS=12,64,256" "PCGC_SKIP_EMPTY=1 PCGC_CHUNKS=16,64,256" "PCGC_SKIP_EMPTY=1 PCGC_CHUNKS=24,64,256" > gpurun_out/r4b/ab.txt 2>&1
PCGC_SKIP_EMPTY=1 python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4b/bench_skip1.json 2>gpurun_out/r4b/bench_skip1.err
PCGC_SKIP_EMPTY=1 PCGC_CHUNKS=16,64,256 python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4b/bench_skip1_c16.json 2>gpurun_out/r4b/bench_skip1_c16.err
tail -4 gpurun_out/r4b/t_skip.txt; grep -a "full cloud vs\|passed\|failed" gpurun_out/r4b/t_golden.txt; cat gpurun_out/r4b/ab.txt
