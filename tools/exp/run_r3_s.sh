mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r3s/pytest.log 2>&1; tail -4 gpurun_out/r3s/pytest.log
for e in "PCGC_DW_MFMA32=1" "PCGC_DW_MFMA32=0" "PCGC_DW_MFMA32=1" "PCGC_DW_MFMA32=0"; do
env $e timeout 300 python tools/bench_train.py 30 2>/dev/null | sed "s/^/$e /"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3s/prof_train -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3s/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/r3s/prof_train -name "*.db" | head -1) > gpurun_out/r3s/train_kernel_stats.csv
rm -rf gpurun_out/r3s/prof_train
grep "dw_" gpurun_out/r3s/train_kernel_stats.csv | cut -c1-150
