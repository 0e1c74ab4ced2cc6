#!/bin/bash
# The three counter passes of tools/gpu_round.sh alone (headline workload only).  Usage: tools/pmc_only.sh <tag>
set -u
TAG=${1:-rXX}; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 bench.py --steps 3 --warmup 3 --no-cpu --no-ref-configs > $OUT/${TAG}_pmc_sq.json 2> $OUT/${TAG}_pmc_sq.err || exit 1
echo PMC_OK
