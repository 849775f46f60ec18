#!/bin/bash
# Round 4: the three hyper rate points the reference lists (eval_ablation_studies.py:71-77) that were missing,
# trained by warm start (the reference's recipe, README.md:86): a0.75b3 and a3.5b3 from a2b3, a16b3 from a10b3.
# usage (on the GPU box): bash tools/run/train_rate_points.sh [minutes per point]
MIN=${1:-9}
OUT=gpurun_out/ckpt_r4
mkdir -p $OUT
python tools/train_ckpt.py --alpha 3.5  --beta 3 --lr 1e-4 --minutes $MIN --init checkpoints/hyper/a2.00b3.00  --out $OUT > $OUT/log_a3.5.txt 2>&1
python tools/train_ckpt.py --alpha 16   --beta 3 --lr 1e-4 --minutes $MIN --init checkpoints/hyper/a10.00b3.00 --out $OUT > $OUT/log_a16.txt 2>&1
python tools/train_ckpt.py --alpha 0.75 --beta 3 --lr 1e-4 --minutes $MIN --init checkpoints/hyper/a2.00b3.00  --out $OUT > $OUT/log_a0.75.txt 2>&1
tail -3 $OUT/log_a3.5.txt $OUT/log_a16.txt $OUT/log_a0.75.txt | cut -c1-600
