#!/bin/bash
# same-box A/B of two source trees (this one against a copy of an older commit built under ab_old/): interleaved rounds of
# tools/exp/t_codec.py.   tools/run/ab_trees.sh OUT ROUNDS [N_PER_ROUND]
OUT=$1; ROUNDS=${2:-3}; N=${3:-30}
mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  (cd ab_old && python tools/exp/t_codec.py $N 2>/dev/null | tail -n 1 | sed 's/^/old: /')
  python tools/exp/t_codec.py $N 2>/dev/null | tail -n 1 | sed 's/^/new: /'
done | tee $OUT/ab_trees.txt
