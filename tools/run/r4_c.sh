#!/bin/bash
# round 4, call C: whole GPU suite + smoke + the round's profiles
mkdir -p gpurun_out/r4c
python -m pytest tests -q -m gpu -x -s > gpurun_out/r4c/gpu_suite.txt 2>&1
tail -5 gpurun_out/r4c/gpu_suite.txt
grep -a "config 3\|frame [0-9]:\|full cloud vs\|empty-space" gpurun_out/r4c/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c/smoke.txt 2>&1; tail -2 gpurun_out/r4c/smoke.txt
bash tools/collect_profiles.sh r04_vA > gpurun_out/r4c/collect.txt 2>&1
tail -12 gpurun_out/r4c/collect.txt
