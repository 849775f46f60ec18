set -x
mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; tail -4 gpurun_out/r3h/pytest.log
timeout 600 python bench.py > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err; tail -c 600 gpurun_out/r3h/bench.json
timeout 300 python tools/host_timeline.py 100 mid > gpurun_out/r3h/timeline_mid.txt 2>&1
timeout 300 python tools/host_timeline.py 100 trained > gpurun_out/r3h/timeline_trained.txt 2>&1
