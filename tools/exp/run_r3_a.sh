set -x
mkdir -p gpurun_out/r3a
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_mfma_overlap.hip -o /tmp/exp_mfma_overlap && timeout 300 /tmp/exp_mfma_overlap > gpurun_out/r3a/mfma_overlap.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3a/test_train.log 2>&1; tail -5 gpurun_out/r3a/test_train.log
PCGC_DW_MFMA=0 timeout 300 python tools/bench_train.py 30 > gpurun_out/r3a/bench_train_valu.txt 2>&1
PCGC_DW_MFMA=1 timeout 300 python tools/bench_train.py 30 > gpurun_out/r3a/bench_train_mfma.txt 2>&1
cat gpurun_out/r3a/bench_train_*.txt
mkdir -p gpurun_out/ckpt
timeout 1700 python tools/train_ckpt.py --alpha 0.75 --lr 4e-4 --minutes 22 --clouds 24 --out gpurun_out/ckpt > gpurun_out/ckpt/a075.log 2>&1; tail -3 gpurun_out/ckpt/a075.log
