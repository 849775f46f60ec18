mkdir -p gpurun_out/r3ad
for e in "A=1" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=24" "PCGC_FIRST_SLICE=32" "PCGC_SLICES=3" "A=1" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=24" "PCGC_FIRST_SLICE=32" "PCGC_SLICES=3"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3ad/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3ad/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
