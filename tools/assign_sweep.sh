#!/bin/bash
# Development probe: E-step time against N around whole rounds of 512 workgroups (131 072 points each), the
# one launch of the plain grid against the product's two launches (whole rounds + tail) and the persistent work-queue variants.   Usage (GPU box): bash tools/assign_sweep.sh <tag>
tag=${1:-sweep}
mkdir -p gpurun_out/$tag
out=gpurun_out/$tag/assign_sweep.txt
: > $out
for n in 1179648 1212416 1245184 1250000 1277952 1310720 1343488; do
  echo "N=$n ($(python3 -c "print(round($n/131072,3))") rounds)" >> $out
  timeout -k 10 60 tools/assign_tune $n 1024 "SPLIT|QUEUE small 2|QUEUE small 0" >> $out 2>&1 || exit 1
done
echo "N=10000000" >> $out
timeout -k 10 100 tools/assign_tune 10000000 1024 "SPLIT|QUEUE" >> $out 2>&1 || exit 1
timeout -k 10 60 tools/assign_trace 1250000 1024 2 >> $out 2>&1 || exit 1
timeout -k 10 60 tools/assign_trace 10000000 1024 2 >> $out 2>&1 || exit 1
timeout -k 10 30 tools/hwid_probe 512 > gpurun_out/$tag/hwid_probe.txt 2>&1 || exit 1
cat $out
