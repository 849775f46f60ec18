#!/bin/bash
# Collect the round's judged evidence on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r02_vB
# -> gpurun_out/<tag>_bench.json                plain `python bench.py --steps 20 --warmup 3`
#    gpurun_out/<tag>_roofline_table.md         every conv kernel of one step (PCGC_BENCH_TOP=40 bench line -> tools/roofline_table.py)
#    gpurun_out/<tag>_kernel_stats_pipes1.csv   rocprofv3 --kernel-trace --stats of bench.py (PCGC_PIPES=1), per-kernel table
#    gpurun_out/<tag>_bench_under_rocprof_pipes1.json   the bench line that traced run printed
#    gpurun_out/<tag>_pmc_per_kernel.csv        FETCH_SIZE / WRITE_SIZE / SQ counters, separate --pmc passes, merged per kernel
#    gpurun_out/<tag>_train_kernel_stats.csv    rocprofv3 --kernel-trace --stats of tools/bench_train.py (one train_hyper step, 8 x 64^3)
#    gpurun_out/<tag>_train_pmc_per_kernel.csv  the same three counter groups for the training step
# The program after `--` is python3 itself (no env / bash hop under the profiler); PMC passes carry --kernel-trace only.
set -u
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
PCGC_BENCH_TOP=40 python bench.py --no-extras --cpu-cubes 0 --steps 10 > $OUT/${TAG}_bench_top40.json 2>> $OUT/${TAG}_bench.err
python tools/roofline_table.py $OUT/${TAG}_bench_top40.json > $OUT/${TAG}_roofline_table.md 2>> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
export PCGC_PIPES=1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o s -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-cubes 0 --no-extras \
  > $OUT/${TAG}_bench_under_rocprof_pipes1.json 2> $OUT/prof_stats.err
python3 $R/tools/rocpd_stats.py $(find $OUT/prof_stats -name "*.db") > $OUT/${TAG}_kernel_stats_pipes1.csv
DBS=""
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/prof_pmc$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-cubes 0 --no-roofline --no-extras \
    > /dev/null 2> $OUT/prof_pmc$i.err
  DBS="$DBS $(find $OUT/prof_pmc$i -name '*.db')"
done
python3 $R/tools/rocpd_pmc.py $DBS > $OUT/${TAG}_pmc_per_kernel.csv
rm -rf $OUT/prof_stats $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3 $OUT/prof_pmc4
# ---- BASELINE configs[3]: the training step
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tstats -o t -- python3 $R/tools/bench_train.py 10 > $OUT/${TAG}_train_under_rocprof.txt 2> $OUT/prof_tstats.err
python3 $R/tools/rocpd_stats.py $(find $OUT/prof_tstats -name "*.db" | head -1) > $OUT/${TAG}_train_kernel_stats.csv
DBS=""
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/prof_tpmc$i -o p -- python3 $R/tools/bench_train.py 3 > /dev/null 2> $OUT/prof_tpmc$i.err
  DBS="$DBS $(find $OUT/prof_tpmc$i -name '*.db')"
done
python3 $R/tools/rocpd_pmc.py $DBS > $OUT/${TAG}_train_pmc_per_kernel.csv
rm -rf $OUT/prof_tstats $OUT/prof_tpmc1 $OUT/prof_tpmc2 $OUT/prof_tpmc3
cd $R
tail -c 400 $OUT/${TAG}_bench.json
head -5 $OUT/${TAG}_kernel_stats_pipes1.csv
head -4 $OUT/${TAG}_pmc_per_kernel.csv
head -4 $OUT/${TAG}_train_pmc_per_kernel.csv | cut -c1-200
