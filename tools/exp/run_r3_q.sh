mkdir -p gpurun_out/r3q
for e in "A=1" "PCGC_SLICES=1 PCGC_FIRST_SLICE=40" "PCGC_SLICES=1 PCGC_FIRST_SLICE=32" "PCGC_SLICES=1 PCGC_FIRST_SLICE=48" "PCGC_PIPES=3" "PCGC_PIPES=3 PCGC_SLICES=1" "A=1" "PCGC_SLICES=1 PCGC_FIRST_SLICE=40" "PCGC_SLICES=1 PCGC_FIRST_SLICE=32" "PCGC_SLICES=1 PCGC_FIRST_SLICE=48" "PCGC_PIPES=3" "PCGC_PIPES=3 PCGC_SLICES=1"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3q/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3q/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
