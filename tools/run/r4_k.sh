#!/bin/bash
mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_parity.py -q -x -k "empty_space or slot_invariant or batch_slot or config5 or transforms" -s > gpurun_out/r4k/t.txt 2>&1; tail -3 gpurun_out/r4k/t.txt; grep -a "empty-space" gpurun_out/r4k/t.txt | cut -c1-200
bash tools/run/ab_env.sh gpurun_out/r4k 4 "-" "PCGC_SKIP_EMPTY=0" > gpurun_out/r4k/ab.txt 2>&1; cat gpurun_out/r4k/ab.txt
python tools/exp/t_file_level.py > gpurun_out/r4k/file_level.txt 2>&1; tail -12 gpurun_out/r4k/file_level.txt
