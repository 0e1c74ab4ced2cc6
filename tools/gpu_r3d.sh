#!/bin/bash
# per-kernel times of the eth fit with one centroid image per restart (group 0) and with restart groups of 4
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for g in 0 4; do
  rm -rf $OUT/fit_prof_g$g
  MTP_RESTART_GROUP=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_prof_g$g -- python3 tools/profile_fit.py eth > $OUT/fit_prof_g$g.log 2>&1 || { tail -20 $OUT/fit_prof_g$g.log; exit 1; }
  tail -2 $OUT/fit_prof_g$g.log
  python3 tools/kstats.py $OUT/fit_prof_g$g > $OUT/r3d_fit_eth_kernel_stats_g$g.txt 2>&1; grep "k_assign\|k_accum\|k_mstep" $OUT/r3d_fit_eth_kernel_stats_g$g.txt
done
echo R3D_OK
