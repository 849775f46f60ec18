mkdir -p gpurun_out/r3r
PCGC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --cpu-cubes 0 > gpurun_out/r3r/bench_n2_gloo.json 2> gpurun_out/r3r/bench_n2.err
tail -c 3000 gpurun_out/r3r/bench_n2_gloo.json
tail -5 gpurun_out/r3r/bench_n2.err
