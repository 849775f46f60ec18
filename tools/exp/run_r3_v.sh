mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py -m gpu -x -q -k "back_to_back or roundtrip_stream or pipelined or scheduler or config5 or sharded or slot or batch" > gpurun_out/r3v/pytest.log 2>&1; tail -4 gpurun_out/r3v/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3v/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3v/b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"
done
timeout 600 python tools/soak.py 160 trained 8 2>&1 | tail -2
