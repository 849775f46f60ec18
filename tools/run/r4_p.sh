#!/bin/bash
bash tools/run/ab_trees.sh gpurun_out/r4p 4 30
