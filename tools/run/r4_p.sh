#!/bin/bash
bash tools/run/ab_codec_env.sh gpurun_out/r4p 2 30 "-" "PCGC_PIPES=3" "PCGC_PIPES=4" "PCGC_DEC_SLICES=2" "PCGC_FIRST_SLICE=16" "PCGC_FIRST_SLICE=32" "PCGC_CHUNKS_S=8,32,256"
