mkdir -p gpurun_out/r3ai
for rep in 1 2 3; do
for e in "PCGC_Z_WORKER=1" "PCGC_Z_WORKER=0" "PCGC_Z_WORKER=1 GPU_MAX_HW_QUEUES=12" "PCGC_Z_WORKER=1 GPU_MAX_HW_QUEUES=16"; do
env $e timeout 300 python bench.py --steps 40 --warmup 5 --no-roofline --cpu-cubes 0 2>/dev/null > gpurun_out/r3ai/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3ai/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'], d['stream_of_clouds']['cubes_per_s'], d['large_cloud']['cubes_per_s'])"
done
done
