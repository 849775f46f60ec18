bash tools/collect_profiles.sh r03_vE > gpurun_out/collect_r03_vE.log 2>&1; tail -2 gpurun_out/collect_r03_vE.log | cut -c1-200
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r03_vE_handover_timeline.txt 2>&1
timeout 600 python tools/timeline2.py 300 trained 1 8 > gpurun_out/r03_vE_large_cloud_timeline.txt 2>&1
