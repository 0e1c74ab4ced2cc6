#!/bin/bash
# Per-step cost at the 8-GPU shard size (N/8 = 1.25M points) on ONE GPU: plain, and with the
# one-rank RCCL communicator forced (tests the collective code path, not xGMI).
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 200 python bench.py --n 1250000 --steps 30 --warmup 3 --no-eval --no-cpu > $OUT/shard_plain.json 2> $OUT/shard_plain.err || { tail -20 $OUT/shard_plain.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --points 1250000 --steps 30 --warmup 3 --no-eval --no-cpu --force-collectives > $OUT/shard_rccl.json 2> $OUT/shard_rccl.err || { tail -20 $OUT/shard_rccl.err; exit 1; }
echo PROBE_OK
