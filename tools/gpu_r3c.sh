#!/bin/bash
# round 3: the restart-grouped E-step
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_batched.py tests/test_gpu_reference_sizes.py -m gpu -x -q --timeout 200 --timeout-method thread > $OUT/r3c_tests.log 2>&1 || { tail -60 $OUT/r3c_tests.log; exit 1; }
tail -5 $OUT/r3c_tests.log
for g in ""; do
  echo "== MTP_RESTART_GROUP=$g"
  if [ -n "$g" ]; then export MTP_RESTART_GROUP=$g; else unset MTP_RESTART_GROUP; fi
  true
done
timeout -k 10 300 python tools/lloyd_group_probe.py > $OUT/r3c_group_probe.txt 2>&1 || { tail -20 $OUT/r3c_group_probe.txt; exit 1; }
cat $OUT/r3c_group_probe.txt
echo R3C_OK
