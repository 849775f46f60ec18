"""Print value / ms_per_step / top kernels of a bench.py JSON line:  python tools/print_bench.py gpurun_out/b.json"""
import json
import sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'])
for k in d['roofline']['top_kernels']:
    print('   ', k)
