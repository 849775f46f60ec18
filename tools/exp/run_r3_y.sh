mkdir -p gpurun_out/r3y
for e in "PCGC_CHUNKS=8,64,256" "PCGC_CHUNKS=4,64,256" "PCGC_CHUNKS=6,64,256" "PCGC_CHUNKS=8,32,256" "PCGC_CHUNKS=8,64,128" "PCGC_CHUNKS=8,64,256" "PCGC_CHUNKS=4,64,256" "PCGC_CHUNKS=6,64,256" "PCGC_CHUNKS=8,32,256" "PCGC_CHUNKS=8,64,128"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3y/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3y/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
