#!/bin/bash
# same-box A/B of environment settings on tools/exp/t_codec.py (round trip medians; lower noise than whole bench runs):
#   tools/run/ab_codec_env.sh OUT ROUNDS N "ENV_A" "ENV_B" ...     ("-" = none)
OUT=$1; ROUNDS=$2; N=$3; shift 3
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for e in "$@"; do
    ev=$e; [ "$e" = "-" ] && ev=""
    env $ev python tools/exp/t_codec.py $N 2>/dev/null | grep "round trip" | sed "s/^/$e: /"
  done
done | tee $OUT/ab_codec_env.txt
