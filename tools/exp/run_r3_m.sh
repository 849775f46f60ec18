set -x
mkdir -p gpurun_out/r3m
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r3m/tl.txt 2>&1
for e in "A=1" "PCGC_Z_FIRST=0" "PCGC_HOST_THREADS=32" "PCGC_SWITCH_INTERVAL_US=200" "PCGC_FIRST_SLICE=24" "A=1" "PCGC_Z_FIRST=0" "PCGC_HOST_THREADS=32" "PCGC_SWITCH_INTERVAL_US=200" "PCGC_FIRST_SLICE=24"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3m/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3m/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3m/pytest.log 2>&1; tail -4 gpurun_out/r3m/pytest.log
