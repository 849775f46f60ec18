set -x
mkdir -p gpurun_out/r3c gpurun_out/ckpt
# 1. the whole GPU suite
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest_gpu.log 2>&1; tail -8 gpurun_out/r3c/pytest_gpu.log
# 2. weight-gradient kernels: matrix-core vs VALU tile kernel, per-kernel
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  PCGC_DW_MFMA=$m rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_train_dw$m -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_train_dw$m.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/r3c/prof_train_dw0 -name "*.db" | head -1) > gpurun_out/r3c/train_kernel_stats_dw_valu.csv || true
python tools/rocpd_stats.py $(find gpurun_out/r3c/prof_train_dw1 -name "*.db" | head -1) > gpurun_out/r3c/train_kernel_stats_dw_mfma.csv || true
rm -rf gpurun_out/r3c/prof_train_dw0 gpurun_out/r3c/prof_train_dw1
# 3. operating points: first decode slice / slices / host threads
bash tools/exp/sweep_host.sh
# 4. host timeline of one step on the trained checkpoint
timeout 300 python tools/host_timeline.py 100 trained > gpurun_out/r3c/host_timeline_trained.txt 2>&1
PCGC_FIRST_SLICE=24 timeout 300 python tools/host_timeline.py 100 trained > gpurun_out/r3c/host_timeline_trained_fs24.txt 2>&1
# 5. the other rate points, warm-started from a6b3 (the reference's recipe: README.md:86 --init_ckpt_dir)
timeout 800 python tools/train_ckpt.py --alpha 2 --lr 1e-4 --minutes 9 --clouds 24 --init checkpoints/hyper/a6.00b3.00 --out gpurun_out/ckpt > gpurun_out/ckpt/a2.log 2>&1; tail -2 gpurun_out/ckpt/a2.log | cut -c1-600
timeout 800 python tools/train_ckpt.py --alpha 10 --lr 1e-4 --minutes 9 --clouds 24 --init checkpoints/hyper/a6.00b3.00 --out gpurun_out/ckpt > gpurun_out/ckpt/a10.log 2>&1; tail -2 gpurun_out/ckpt/a10.log | cut -c1-600
