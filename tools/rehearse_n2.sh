#!/bin/bash
# Rehearsal of bench.py's N>1 control flow on ONE GPU: 2 ranks share the device, gloo instead of RCCL
# (the gather is staged through host memory).  Small blocks; not a measurement.
export MASTER_ADDR=127.0.0.1
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 3 --warmup 1 --block-mib 16 --backend gloo
