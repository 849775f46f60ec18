mkdir -p gpurun_out/r3aq
for rep in 1 2 3 4 5 6; do
for e in "PCGC_DEC_SLICES=2" "PCGC_DEC_SLICES=1"; do
env $e timeout 300 python bench.py --steps 60 --warmup 5 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3aq/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3aq/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
done
