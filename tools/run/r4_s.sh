#!/bin/bash
mkdir -p gpurun_out/r4s
python -m pytest tests/test_gpu_parity.py -q -x -k "cli or streamed_postprocess or file or config5 or rd_harness" > gpurun_out/r4s/t.txt 2>&1; tail -n 3 gpurun_out/r4s/t.txt
for i in 1 2; do
python tools/exp/prof_cli.py 2>/dev/null | grep "cubes/s" | sed 's/^/new: /'
(cd ab_old && cp ../tools/exp/prof_cli.py tools/exp/prof_cli.py && python tools/exp/prof_cli.py 2>/dev/null | grep "cubes/s" | sed 's/^/HEAD~: /')
done
