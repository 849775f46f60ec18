set -x
mkdir -p gpurun_out/r3k
timeout 300 python tools/timeline2.py 80 trained 2 > gpurun_out/r3k/tl.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "roundtrip_stream or pipelined or scheduler or cross_decode" > gpurun_out/r3k/pytest.log 2>&1; tail -4 gpurun_out/r3k/pytest.log
for e in "PCGC_COPY_STREAM=1" "PCGC_COPY_STREAM=0" "PCGC_COPY_STREAM=1 PCGC_HOST_THREADS=64" "PCGC_COPY_STREAM=1" "PCGC_COPY_STREAM=0" "PCGC_COPY_STREAM=1 PCGC_HOST_THREADS=64"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3k/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3k/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
