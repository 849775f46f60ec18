set -x
mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or config3 or scheduler or roundtrip_stream" > gpurun_out/r3p/pytest.log 2>&1; tail -3 gpurun_out/r3p/pytest.log
timeout 600 python bench.py --cpu-cubes 0 > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r3p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('stream_of_clouds'), d.get('large_cloud'))
for o in d['operating_points']: print(o)"
