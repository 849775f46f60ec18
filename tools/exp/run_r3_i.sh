set -x
mkdir -p gpurun_out/r3i
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "roundtrip_stream or pipelined or scheduler or cross_decode or config3" > gpurun_out/r3i/pytest.log 2>&1; tail -4 gpurun_out/r3i/pytest.log
for e in 1 0 1 0; do
PCGC_EARLY_RANGES=$e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3i/bench_early$e.json
python -c "
import json
d=json.loads(open('gpurun_out/r3i/bench_early$e.json').read().strip().splitlines()[-1])
print('early=$e', d['value'], d['ms_per_step'])"
done
timeout 600 python bench.py --cpu-cubes 0 --no-roofline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('stream_of_clouds'))"
