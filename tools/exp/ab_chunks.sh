#!/bin/bash
# interleaved A/B of PCGC_CHUNKS on one box: value ms_per_step per run
python bench.py --no-extras --cpu-cubes 0 --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
  for c in 8,64,256 12,64,256 16,64,256 10,64,256; do
    PCGC_CHUNKS=$c python bench.py --no-extras --cpu-cubes 0 --no-roofline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'])"
  done
done
