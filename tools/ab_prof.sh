#!/bin/bash
# Development A/B: per-kernel times of the batched seeding under rocprofv3 for every library under tools/ab/ and the tree's
# own (run on the GPU box's copy of the tree only: it overwrites pccs_accel/libmtp_accel.so).
set -u
export TMPDIR=/tmp
L=modular-trajectory-prediction-toolkit_amd/pccs_accel/libmtp_accel.so
cp $L /tmp/libmtp_new.so
for f in tools/ab/*.so /tmp/libmtp_new.so; do
  cp $f $L
  name=$(basename $f .so)
  rm -rf /tmp/prof_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 tools/seed_probe.py > /tmp/prof_$name.log 2>&1 || { tail -5 /tmp/prof_$name.log; exit 1; }
  echo "== $name"; python3 tools/kstats.py /tmp/prof_$name | sort -k3 -n -r | head -8
done
cp /tmp/libmtp_new.so $L
