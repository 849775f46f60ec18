#!/bin/bash
# round 4, call H: kernel trace of the training step in Q4 mode; decoder-slice knobs with the faster analysis
mkdir -p gpurun_out/r4h
R=$(pwd)
python -m pytest tests/test_gpu_train.py -q -x > gpurun_out/r4h/t_train_all.txt 2>&1; tail -3 gpurun_out/r4h/t_train_all.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4h/prof_t -o t -- python3 $R/tools/bench_train.py 10 > $R/gpurun_out/r4h/train_under_rocprof.txt 2> $R/gpurun_out/r4h/prof_t.err
python3 $R/tools/rocpd_stats.py $(find $R/gpurun_out/r4h/prof_t -name "*.db" | head -1) > $R/gpurun_out/r4h/train_q4_kernel_stats.csv
rm -rf $R/gpurun_out/r4h/prof_t
cd $R
head -24 gpurun_out/r4h/train_q4_kernel_stats.csv | cut -c1-150
bash tools/run/ab_env.sh gpurun_out/r4h 2 "-" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=32" "PCGC_FIRST_SLICE=48" "PCGC_DEC_SLICES=2" "PCGC_SLICES=3" > gpurun_out/r4h/ab.txt 2>&1
cat gpurun_out/r4h/ab.txt
