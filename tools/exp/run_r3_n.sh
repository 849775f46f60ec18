set -x
mkdir -p gpurun_out/r3n
for e in "A=1" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=24" "PCGC_FIRST_SLICE=32" "A=1" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=24" "PCGC_FIRST_SLICE=32"; do
env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3n/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3n/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
timeout 600 python bench.py --cpu-cubes 0 > gpurun_out/r3n/bench.json 2> gpurun_out/r3n/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r3n/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('stream_of_clouds'))
for o in d['operating_points']: print(o)
print(d['roofline']['all_conv_tflops'], d['roofline']['frac'], d['train'])"
timeout 300 python tools/timeline2.py 80 mid 1 > gpurun_out/r3n/tl_mid.txt 2>&1
