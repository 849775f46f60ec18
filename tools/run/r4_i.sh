#!/bin/bash
# round 4, call I: encoder head / tail split
mkdir -p gpurun_out/r4i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py -q -x -k "not config3 and not config5" > gpurun_out/r4i/t.txt 2>&1; tail -3 gpurun_out/r4i/t.txt
bash tools/run/ab_env.sh gpurun_out/r4i 3 "PCGC_ENC_TAIL=0" "PCGC_ENC_TAIL=16" "PCGC_ENC_TAIL=24" "PCGC_ENC_TAIL=40" > gpurun_out/r4i/ab.txt 2>&1
cat gpurun_out/r4i/ab.txt
