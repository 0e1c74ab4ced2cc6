#!/bin/bash
# Development A/B: the headline Lloyd step (bench.py, no CPU legs) with every library under tools/ab/ swapped in, then the
# tree's own (run on the GPU box's copy of the tree only: it overwrites pccs_accel/libmtp_accel.so).
# Usage: tools/ab_bench.sh [extra bench.py flags]
set -u
L=modular-trajectory-prediction-toolkit_amd/pccs_accel/libmtp_accel.so
cp $L /tmp/libmtp_new.so
for r in 1 2; do
  for f in tools/ab/*.so /tmp/libmtp_new.so; do
    cp $f $L
    timeout -k 10 300 python bench.py --no-cpu --no-ref-configs --no-eval --no-convergence "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
b = r['ms_breakdown']
print('%-28s step %.3f ms  assign %.3f  accumulate %.3f  rest %.3f' % ('$f'[-28:], r['ms_per_step'], b['assign'], b['accumulate'], b['allreduce+mstep']))
" || exit 1
  done
done
cp /tmp/libmtp_new.so $L
