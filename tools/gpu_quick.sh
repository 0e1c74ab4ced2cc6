#!/bin/bash
# Quick GPU pass: GPU tests, then the bench line (no profiles).  Usage: tools/gpu_quick.sh <tag> [pytest -k expr]
set -u
TAG=${1:-q}
KEXPR=${2:-}
OUT=gpurun_out
mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$KEXPR" > $OUT/${TAG}_gpu_tests.log 2>&1 || { tail -40 $OUT/${TAG}_gpu_tests.log; exit 1; }
else
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/${TAG}_gpu_tests.log 2>&1 || { tail -40 $OUT/${TAG}_gpu_tests.log; exit 1; }
fi
tail -3 $OUT/${TAG}_gpu_tests.log
timeout -k 10 600 python bench.py --no-cpu > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 1; }
echo QUICK_OK
