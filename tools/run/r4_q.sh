#!/bin/bash
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/r4q/t.txt 2>&1; tail -n 3 gpurun_out/r4q/t.txt
for r in 1 2 3; do
  for e in "PCGC_TRAIN_DW_STREAM=0" "PCGC_TRAIN_DW_STREAM=1"; do env $e python tools/exp/t_train_step.py 30 2>/dev/null | tail -n 1 | sed "s/^/$e: /"; done
done | tee gpurun_out/r4q/ab.txt
