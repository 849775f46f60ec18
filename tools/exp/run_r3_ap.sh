R=$(pwd); OUT=$R/gpurun_out; TAG=r03_vF
cd /tmp && export TMPDIR=/tmp
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
cat $OUT/${TAG}_train_under_rocprof.txt
timeout 300 python tools/bench_train.py 30 2>/dev/null
