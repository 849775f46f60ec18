mkdir -p gpurun_out/r3ab
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3ab/pytest.log 2>&1; tail -3 gpurun_out/r3ab/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --cpu-cubes 0 2>/dev/null > gpurun_out/r3ab/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3ab/b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stream_of_clouds']['cubes_per_s'], d['large_cloud']['cubes_per_s'], d['file_level']['cubes_per_s'], d['train']['ms_per_step'])"
done
