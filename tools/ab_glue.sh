#!/bin/bash
# Development A/B: decode / gather / best-of-K glue timings (tests/perf_glue.py) with every library under tools/ab/ swapped in,
# then the tree's own (run on the GPU box's copy of the tree only: it overwrites pccs_accel/libmtp_accel.so).
set -u
L=modular-trajectory-prediction-toolkit_amd/pccs_accel/libmtp_accel.so
cp $L /tmp/libmtp_new.so
for f in tools/ab/*.so /tmp/libmtp_new.so; do
  cp $f $L
  echo "== $f"
  timeout -k 10 300 python tests/perf_glue.py 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    r = json.loads(line)
    print('  %-70s %8.3f ms %s' % (r['kernel'][:70], r['ms'], ('%.0f GB/s' % r['GB/s']) if 'GB/s' in r else ''))
"
done
cp /tmp/libmtp_new.so $L
