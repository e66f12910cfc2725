#!/bin/bash
# One-rank rehearsal of the N > 1 code path on a one-GPU box: RCCL process group, bucketed all-reduce from the backward hooks on
# the side stream, barriers, then the reduce-scatter / all-gather variant.  Not a measurement.
set -e
export MI_FORCE_COMM=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "" "--shard-optimizer"; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 4 --warmup 2 --batch 8 --no-cpu-baseline --no-fp32-line --no-roofline $extra
done
