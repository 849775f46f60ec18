set -x
mkdir -p gpurun_out/pilot
timeout 420 python tools/train_ckpt.py --alpha 6 --lr 1e-4 --minutes 3 --clouds 8 --out gpurun_out/pilot/lr1e-4 > gpurun_out/pilot/lr1e-4.log 2>&1
timeout 420 python tools/train_ckpt.py --alpha 6 --lr 4e-4 --minutes 3 --clouds 8 --out gpurun_out/pilot/lr4e-4 > gpurun_out/pilot/lr4e-4.log 2>&1
tail -5 gpurun_out/pilot/*.log
