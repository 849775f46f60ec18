# operating-point sweeps on the GPU box: first decode slice, entropy slices per pipeline, host coder threads
mkdir -p gpurun_out/sweep
run() {  # label, env...
  label="$1"; shift
  env "$@" timeout 300 python bench.py --profile $prof --steps 10 --warmup 3 --no-roofline --no-extras --cpu-cubes 0 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$prof $label', d['value'], d['ms_per_step'], d['config']['bytes_per_cube'])" | tee -a gpurun_out/sweep/host_sweep.txt
}
for prof in trained sparse mid; do
  for fs in 0 16 24 32 48; do run "first_slice=$fs" PCGC_FIRST_SLICE=$fs; done
  for sl in 3 4; do run "first_slice=24 slices=$sl" PCGC_FIRST_SLICE=24 PCGC_SLICES=$sl; done
  for th in 16 64; do run "first_slice=24 threads=$th" PCGC_FIRST_SLICE=24 PCGC_HOST_THREADS=$th; done
done
