#!/bin/bash
# Development A/B: best-of-K timing with every library under tools/ab/ swapped in, then the tree's own, each with the
# persistent grid (default) and with one workgroup per group of agents
# (run on the GPU box's copy of the tree only: it overwrites pccs_accel/libmtp_accel.so).
set -u
L=modular-trajectory-prediction-toolkit_amd/pccs_accel/libmtp_accel.so
cp $L /tmp/libmtp_new.so
for f in tools/ab/*.so /tmp/libmtp_new.so; do
  cp $f $L
  echo "$f: $(timeout -k 10 200 python tools/bok_probe.py 2>/dev/null | cut -d: -f2 | cut -c1-28 | tr '\n' '|')"
  echo "$f grid=n_groups: $(MTP_BOK_GRID=100000000 timeout -k 10 200 python tools/bok_probe.py 2>/dev/null | cut -d: -f2 | cut -c1-28 | tr '\n' '|')"
done
cp /tmp/libmtp_new.so $L
