mkdir -p gpurun_out/r3ab
for e in "GPU_MAX_HW_QUEUES=4" "A=1" "GPU_MAX_HW_QUEUES=4" "A=1" "GPU_MAX_HW_QUEUES=4" "A=1"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --cpu-cubes 0 2>/dev/null > gpurun_out/r3ab/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3ab/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'], d['stream_of_clouds']['cubes_per_s'], d['large_cloud']['cubes_per_s'])"
done
