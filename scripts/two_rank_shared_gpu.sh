#!/bin/bash
# Dev helper: bench.py with 2 gloo ranks sharing the single GPU, in redundant and leader host modes.
for lm in 0 1; do
  DPMM_LEADER_MODE=$lm DPMM_BENCH_BACKEND=gloo DPMM_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 2972$lm bench.py --gpus 2 --points 1e6 --steps 20 --warmup 2 2>&1 | tail -1 > /tmp/out_$lm.json
  python -c "import json; d=json.load(open('/tmp/out_$lm.json')); print('leader_mode=$lm', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms/step', {k: round(v,3) for k,v in d['host_ms_per_step'].items()})"
done
