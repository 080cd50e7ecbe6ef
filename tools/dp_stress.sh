#!/bin/bash
# usage: tools/dp_stress.sh <label> <steps> [ENV=VAL ...]   -- 2 ranks on one GPU over gloo (tools/dp_stress_worker.py)
label=$1; steps=$2; shift 2
port=$((29400 + RANDOM % 300))
echo "=== $label ($*)"
env HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NPROC:-2} --master-addr 127.0.0.1 --master-port $port tools/dp_stress_worker.py ${SHAPE:-32,48,32} $steps 2>&1 | grep -E "^rank|Error|error" 
