#!/bin/bash
# A/B of environment settings on the headline bench, interleaved (box clocks drift): tools/run/ab_env.sh OUT REPS "ENV_A" "ENV_B" ...
# each ENV_x is a space-separated list of VAR=value (or "-" for none); prints value / ms_per_step per run
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for e in "$@"; do
    ev=$e; [ "$e" = "-" ] && ev=""
    env $ev timeout 600 python bench.py --steps 30 --warmup 4 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null > $OUT/b.json
    python - "$e" $OUT/b.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-40s %8.1f cubes/s %7.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  done
done
