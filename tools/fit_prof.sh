#!/bin/bash
# Per-kernel times of one reference-sized batched fit (eth) under rocprofv3.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/fit_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_prof -- python3 tools/profile_fit.py ${FIT_CFG:-eth} > $OUT/fit_prof.log 2>&1 || { tail -20 $OUT/fit_prof.log; exit 1; }
tail -12 $OUT/fit_prof.log
python3 tools/kstats.py $OUT/fit_prof > $OUT/fit_eth_kernel_stats.txt 2>&1; head -20 $OUT/fit_eth_kernel_stats.txt
