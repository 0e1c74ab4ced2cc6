#!/bin/bash
# Development A/B: seeding time with every library under tools/ab/ swapped in, then the tree's own
# (run on the GPU box's copy of the tree only: it overwrites pccs_accel/libmtp_accel.so).
set -u
L=modular-trajectory-prediction-toolkit_amd/pccs_accel/libmtp_accel.so
cp $L /tmp/libmtp_new.so
for r in 1 2; do
  for f in tools/ab/*.so /tmp/libmtp_new.so; do
    cp $f $L
    echo "$f:"; timeout -k 10 200 python tools/seed_probe.py 2>/dev/null || exit 1
  done
done
cp /tmp/libmtp_new.so $L
