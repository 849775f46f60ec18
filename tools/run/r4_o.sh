#!/bin/bash
mkdir -p gpurun_out/r4o
python tools/exp/t_synth_chunks.py > gpurun_out/r4o/synth.txt 2>&1
