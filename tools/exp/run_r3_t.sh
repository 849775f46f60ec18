mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r3t/pytest.log 2>&1; tail -6 gpurun_out/r3t/pytest.log
for e in "PCGC_DW_PAIR=1" "PCGC_DW_PAIR=0" "PCGC_DW_PAIR=1" "PCGC_DW_PAIR=0"; do
env $e timeout 300 python tools/bench_train.py 30 2>/dev/null | sed "s/^/$e /"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3t/prof_train -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3t/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/r3t/prof_train -name "*.db" | head -1) > gpurun_out/r3t/train_kernel_stats.csv
rm -rf gpurun_out/r3t/prof_train
grep "dw_" gpurun_out/r3t/train_kernel_stats.csv | cut -c1-150 | head -12
