#!/bin/bash
# One GPU-box pass: GPU tests, the bench line, the rocprofv3 kernel summary of the same command and
# the PMC passes (one counter block per run).  Usage: tools/gpu_round.sh <tag>   (outputs: gpurun_out/<tag>_*)
set -u
TAG=${1:-rXX}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/${TAG}_gpu_tests.log 2>&1 || { tail -30 $OUT/${TAG}_gpu_tests.log; exit 1; }
tail -3 $OUT/${TAG}_gpu_tests.log
timeout -k 10 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-convergence --no-ref-configs > $OUT/${TAG}_bench_prof.json 2> $OUT/${TAG}_bench_prof.err || { tail -20 $OUT/${TAG}_bench_prof.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs --no-convergence > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs --no-convergence > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs --no-convergence > $OUT/${TAG}_pmc_sq.json 2> $OUT/${TAG}_pmc_sq.err || exit 1
echo ALL_OK
