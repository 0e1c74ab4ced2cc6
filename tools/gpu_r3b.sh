#!/bin/bash
# round 3: the sklearn one-thread comparison mode
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_reference_sizes.py -m gpu -x -q --timeout 300 --timeout-method thread -k "one_thread or pairwise" --durations=8 > $OUT/r3b_1t_tests.log 2>&1 || { tail -60 $OUT/r3b_1t_tests.log; exit 1; }
tail -15 $OUT/r3b_1t_tests.log
echo R3B_OK
