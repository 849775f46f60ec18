set -x
mkdir -p gpurun_out/r3l
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r3l/tl.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py tests/test_trained_checkpoints.py -m gpu -x -q > gpurun_out/r3l/pytest.log 2>&1; tail -4 gpurun_out/r3l/pytest.log
for e in "A=1" "PCGC_ENTROPY_STREAM=0" "PCGC_Z_FIRST=1" "PCGC_SWITCH_INTERVAL_US=200" "PCGC_HOST_THREADS=64" "A=1" "PCGC_ENTROPY_STREAM=0" "PCGC_Z_FIRST=1" "PCGC_SWITCH_INTERVAL_US=200" "PCGC_HOST_THREADS=64"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3l/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3l/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
