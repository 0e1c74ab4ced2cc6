#!/bin/bash
# Idle time between Lloyd iterations at the 8-GPU shard size (N/8 = 1.25M points) on ONE GPU, with a one-rank RCCL
# communicator so that every collective of the N > 1 path is issued: kernel trace of the library loop
# (mtp_lloyd_sharded_f32) and of the step-level Python driver.  Usage: tools/gap_probe.sh <tag>
set -u
TAG=${1:-gap}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for LOOP in library python; do
  rm -rf $OUT/trace_$LOOP
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$LOOP -- python3 bench.py --gpus 1 --points 1250000 --steps 30 --warmup 3 --no-eval --no-cpu --no-ref-configs --force-collectives --loop $LOOP > $OUT/bench_$LOOP.json 2> $OUT/bench_$LOOP.err || { tail -20 $OUT/bench_$LOOP.err; exit 1; }
  echo "== loop: $LOOP" | tee -a $OUT/gaps.txt
  python3 tools/step_gaps.py $OUT/trace_$LOOP | tee -a $OUT/gaps.txt
  python3 -c "import json,sys; j=json.load(open('$OUT/bench_$LOOP.json')); print(j['loop'], 'ms_per_step', round(j['ms_per_step'],4), j['ms_breakdown'])" | tee -a $OUT/gaps.txt
done
