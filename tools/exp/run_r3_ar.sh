bash tools/collect_profiles.sh r03_vG > gpurun_out/collect_r03_vG.log 2>&1; tail -2 gpurun_out/collect_r03_vG.log | cut -c1-200
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r03_vG_handover_timeline.txt 2>&1
timeout 600 python tools/timeline2.py 300 trained 1 8 > gpurun_out/r03_vG_large_cloud_timeline.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03_vG_gpu_suite.txt 2>&1; tail -2 gpurun_out/r03_vG_gpu_suite.txt
