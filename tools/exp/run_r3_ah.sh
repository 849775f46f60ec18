timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r03_vD.log 2>&1; tail -3 gpurun_out/pytest_gpu_r03_vD.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_profiles.sh r03_vD > gpurun_out/collect_r03_vD.log 2>&1; tail -3 gpurun_out/collect_r03_vD.log | cut -c1-200
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r03_vD_handover_timeline.txt 2>&1
timeout 600 python tools/soak.py 200 trained 10 2>&1 | tail -1
