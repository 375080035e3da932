#!/bin/bash
# the driver's multi-rank launch style on the one GPU (ranks share it: gloo): default workload
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/y
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --workspace-gib 48 ) > gpurun_out/y/bench_torchrun2.json 2> gpurun_out/y/bench_torchrun2.err
echo "rc=$?"; tail -c 1500 gpurun_out/y/bench_torchrun2.json; echo; tail -3 gpurun_out/y/bench_torchrun2.err
