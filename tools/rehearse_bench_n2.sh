#!/bin/bash
# The N > 1 flow of bench.py (rendezvous, shards, barriers, max over ranks, one JSON line from rank 0) rehearsed on a ONE-GPU
# box: 2 or 3 ranks on device 0 over gloo (RCCL refuses two ranks on one device).  Not a measurement: it only shows that the
# script's multi-rank path runs end to end on hardware.     tools/rehearse_bench_n2.sh [ranks] [rows] [more bench.py arguments]
N=${1:-2}; ROWS=${2:-1500000}
cd "$(dirname "$0")/.."
RBL_BENCH_REHEARSE_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29741 \
    bench.py --gpus $N --steps 20 --warmup 5 --rows $ROWS "${@:3}"
