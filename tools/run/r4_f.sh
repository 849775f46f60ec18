#!/bin/bash
# round 4, call F: skipping extended to down_1 + the 32^3 stage; then the a6b3 checkpoint re-derived by warm start from a10b3
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or slot_invariant or batch_slot" -s > gpurun_out/r4f/t_skip.txt 2>&1
tail -4 gpurun_out/r4f/t_skip.txt; grep -a "empty-space" gpurun_out/r4f/t_skip.txt
bash tools/run/ab_env.sh gpurun_out/r4f 3 "PCGC_SKIP_EMPTY=0" "PCGC_SKIP_MID=0" "-" "PCGC_CHUNKS_A=24,64,256" "PCGC_CHUNKS_A=32,64,256" > gpurun_out/r4f/ab.txt 2>&1
cat gpurun_out/r4f/ab.txt
python bench.py --steps 20 --warmup 3 --no-extras --cpu-cubes 0 > gpurun_out/r4f/bench.json 2>gpurun_out/r4f/bench.err
OUT=gpurun_out/ckpt_r4b
mkdir -p $OUT
python tools/train_ckpt.py --alpha 6 --beta 3 --lr 1e-4 --minutes 9 --init checkpoints/hyper/a10.00b3.00 --out $OUT > $OUT/log_a6.txt 2>&1
tail -2 $OUT/log_a6.txt | cut -c1-600
