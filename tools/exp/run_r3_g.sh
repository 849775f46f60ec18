set -x
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r3g/pytest.log 2>&1; tail -4 gpurun_out/r3g/pytest.log
PCGC_DW_MFMA=1 timeout 300 python tools/bench_train.py 30 2>/dev/null | tee gpurun_out/r3g/bench_train_mfma.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_train -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/r3g/prof_train -name "*.db" | head -1) > gpurun_out/r3g/r03_vD_train_kernel_stats.csv
rm -rf gpurun_out/r3g/prof_train
head -16 gpurun_out/r3g/r03_vD_train_kernel_stats.csv | cut -c1-140
