set -x
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py -m gpu -x -q > gpurun_out/r3o/pytest.log 2>&1; tail -3 gpurun_out/r3o/pytest.log
for p in mid sparse trained mid sparse trained; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 --profile $p 2>/dev/null > gpurun_out/r3o/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3o/b.json').read().strip().splitlines()[-1])
print('$p', d['value'], d['ms_per_step'])"
done
