timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r03_vC.log 2>&1; tail -3 gpurun_out/pytest_gpu_r03_vC.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh r03_vC > gpurun_out/collect_r03_vC.log 2>&1; tail -3 gpurun_out/collect_r03_vC.log | cut -c1-300
