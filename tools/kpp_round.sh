#!/bin/bash
# k-means++ kernel check: instruction order probe, parity tests touching the seeding, reference-sized fits.
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 60 tools/mfma_f64_order > $OUT/mfma_f64_order.txt 2>&1; echo "probe rc=$?"; cat $OUT/mfma_f64_order.txt
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "seeding or kmeanspp or plusplus or batched or seeded or cluster_data or xoe or fit" > $OUT/kpp_tests.log 2>&1 || { tail -40 $OUT/kpp_tests.log; exit 1; }
tail -3 $OUT/kpp_tests.log
timeout -k 10 600 python tools/bench_configs.py --configs eth,sdd > $OUT/kpp_configs.log 2>&1 || { tail -20 $OUT/kpp_configs.log; exit 1; }
cat $OUT/kpp_configs.log
