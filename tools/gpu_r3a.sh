#!/bin/bash
# round 3, first pass: the new multi-process tests, the relocation / merge tests, then the whole suite + rehearsals
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_dist_stub.py tests/test_gpu_parity.py -m gpu -x -q --timeout 150 --timeout-method thread -k "stub or far_merge or relocation or in_library" > $OUT/r3a_new_tests.log 2>&1 || { tail -40 $OUT/r3a_new_tests.log; exit 1; }
tail -3 $OUT/r3a_new_tests.log
MTP_RCCL_LIB=$PWD/tests/stub_nccl/libstubnccl.so timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --points 2000000 --no-cpu --no-ref-configs --eval-n 100000 > $OUT/r3a_bench_stub2.json 2> $OUT/r3a_bench_stub2.err || { tail -30 $OUT/r3a_bench_stub2.err; exit 1; }
cat $OUT/r3a_bench_stub2.json | head -c 1500; echo
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --points 200000 --no-cpu --no-ref-configs --eval-n 100000 > $OUT/r3a_bench_gloo2.json 2> $OUT/r3a_bench_gloo2.err || { tail -30 $OUT/r3a_bench_gloo2.err; exit 1; }
cat $OUT/r3a_bench_gloo2.json | head -c 600; echo
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/r3a_gpu_tests.log 2>&1 || { tail -40 $OUT/r3a_gpu_tests.log; exit 1; }
tail -3 $OUT/r3a_gpu_tests.log
echo R3A_OK
