set -x
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; tail -4 gpurun_out/r3f/pytest.log
PCGC_DW_MFMA=1 timeout 300 python tools/bench_train.py 30 2>/dev/null | tee gpurun_out/r3f/bench_train_mfma.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3f/prof_train -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3f/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/r3f/prof_train -name "*.db" | head -1) > gpurun_out/r3f/r03_vC_train_kernel_stats.csv
rm -rf gpurun_out/r3f/prof_train
grep "16x4" gpurun_out/r3f/r03_vC_train_kernel_stats.csv | cut -c1-140
for r in 0 1 0 1; do
  PCGC_XCD_REMAP_OUT=$r PCGC_BENCH_TOP=40 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --cpu-cubes 0 2>/dev/null > gpurun_out/r3f/bench_remap$r.json
  python -c "
import json
d=json.loads(open('gpurun_out/r3f/bench_remap$r.json').read().strip().splitlines()[-1])
k=[t for t in d['roofline']['top_kernels'] if 'deconv_out' in t['kernel']]
print('remap=$r', d['value'], d['ms_per_step'], k)"
done
