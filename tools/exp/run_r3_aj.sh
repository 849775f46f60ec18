mkdir -p gpurun_out/r3aj
python - <<'PY'
import torch, glob, os
pr = torch.cuda.get_device_properties(0)
addr = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
for d in glob.glob("/sys/class/drm/card*/device"):
    if os.path.basename(os.path.realpath(d)).lower().startswith(addr):
        print("gpu", addr, d, open(d + "/numa_node").read().strip(), open(d + "/local_cpulist").read().strip())
PY
for rep in 1 2 3; do
for e in "none" "0-63,128-191" "64-127,192-255"; do
if [ "$e" = "none" ]; then P=""; else P="taskset -c $e"; fi
$P timeout 300 python bench.py --steps 40 --warmup 5 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3aj/b.json
python -c "
import json
d=json.loads(open('gpurun_out/r3aj/b.json').read().strip().splitlines()[-1])
print('$e', d['value'], d['ms_per_step'])"
done
done
