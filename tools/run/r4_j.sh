#!/bin/bash
# round 4, call J: the whole GPU suite, smoke, soak, hand-over timeline, the round's profiles (tools/collect_profiles.sh)
mkdir -p gpurun_out/r4j
python -m pytest tests -q -m gpu -x > gpurun_out/r4j/gpu_suite.txt 2>&1; tail -4 gpurun_out/r4j/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4j/smoke.txt 2>&1; tail -1 gpurun_out/r4j/smoke.txt
python tools/soak.py 200 trained 20 > gpurun_out/r04_vB_soak.txt 2>&1; tail -2 gpurun_out/r04_vB_soak.txt
python tools/timeline2.py 40 trained 3 > gpurun_out/r04_vB_handover_timeline.txt 2>&1; grep -a "^step" gpurun_out/r04_vB_handover_timeline.txt
bash tools/collect_profiles.sh r04_vB > gpurun_out/r4j/collect.txt 2>&1
tail -14 gpurun_out/r4j/collect.txt | cut -c1-300
