mkdir -p gpurun_out/r3aa
for e in "A=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "A=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --cpu-cubes 0 2>/dev/null > gpurun_out/r3aa/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3aa/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'], d['stream_of_clouds']['cubes_per_s'], d['large_cloud']['cubes_per_s'], d['train']['ms_per_step'])"
done
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/timeline2.py 300 trained 1 8 > gpurun_out/r3aa/tl_large_q8.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/timeline2.py 80 trained 2 > gpurun_out/r3aa/tl_q8.txt 2>&1
