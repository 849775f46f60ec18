#!/bin/bash
# round 4, call A: new parity tests + empty-space skipping A/B
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or slot_invariant" -s 2>&1 | tail -15 > gpurun_out/r4a/t_skip.txt
python -m pytest tests/test_trained_checkpoints.py -q -x -m gpu -s 2>&1 | tail -15 > gpurun_out/r4a/t_trained.txt
python -m pytest tests/test_gpu_sharding.py -q -x 2>&1 | tail -8 > gpurun_out/r4a/t_shard.txt
bash tools/run/ab_env.sh gpurun_out/r4a 3 "PCGC_SKIP_EMPTY=0" "PCGC_SKIP_EMPTY=1" "PCGC_SKIP_EMPTY=1 PCGC_CHUNKS=16,64,256" > gpurun_out/r4a/ab.txt 2>&1
PCGC_SKIP_EMPTY=1 python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4a/bench_skip1.json 2>gpurun_out/r4a/bench_skip1.err
PCGC_SKIP_EMPTY=0 python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4a/bench_skip0.json 2>gpurun_out/r4a/bench_skip0.err
cat gpurun_out/r4a/t_skip.txt gpurun_out/r4a/t_trained.txt gpurun_out/r4a/t_shard.txt gpurun_out/r4a/ab.txt
