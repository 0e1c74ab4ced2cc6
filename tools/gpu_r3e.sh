#!/bin/bash
# round 3: per-D accumulate kernels -- parity tests that touch the sums, then the eth profile and the headline step
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_batched.py tests/test_gpu_reference_sizes.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/r3e_tests.log 2>&1 || { tail -40 $OUT/r3e_tests.log; exit 1; }
tail -3 $OUT/r3e_tests.log
rm -rf $OUT/fit_prof_e
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fit_prof_e -- python3 tools/profile_fit.py eth > $OUT/fit_prof_e.log 2>&1 || { tail -20 $OUT/fit_prof_e.log; exit 1; }
python3 tools/kstats.py $OUT/fit_prof_e > $OUT/r3e_fit_eth_kernel_stats.txt 2>&1; grep "k_assign\|k_accum\|k_mstep" $OUT/r3e_fit_eth_kernel_stats.txt
timeout -k 10 300 python bench.py --no-cpu --no-ref-configs --no-convergence --no-eval > $OUT/r3e_bench.json 2> $OUT/r3e_bench.err || { tail -20 $OUT/r3e_bench.err; exit 1; }
python3 -c "import json; j=json.load(open('$OUT/r3e_bench.json')); print(j['ms_per_step'], j['ms_breakdown'])"
timeout -k 10 300 python bench.py --no-cpu --no-ref-configs --no-convergence --no-eval --points 1250000 > $OUT/r3e_bench_shard.json 2> $OUT/r3e_bench_shard.err || { tail -20 $OUT/r3e_bench_shard.err; exit 1; }
python3 -c "import json; j=json.load(open('$OUT/r3e_bench_shard.json')); print(j['ms_per_step'], j['ms_breakdown'])"
echo R3E_OK
