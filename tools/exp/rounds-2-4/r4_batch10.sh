#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b10
mkdir -p $OUT
cd $R
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29679 bench.py --gpus 8 --backend gloo --size 128 --steps 2 --warmup 1 > $OUT/bench8.out 2> $OUT/bench8.err
echo "rc $?"
grep -v "Gloo\] Rank" $OUT/bench8.err | grep -B2 -A25 "Traceback\|Error\|error" | head -120
tail -c 1500 $OUT/bench8.out
