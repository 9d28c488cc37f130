#!/bin/bash
# Dev helper: bench.py with EIGHT gloo ranks sharing the single GPU -- not a scaling measurement (the GPU work is serialised),
# but it exposes host-side contention of an 8-rank run under the box's 16-CPU quota (spinning syncs + the leader's pool).
for lm in 1; do
  DPMM_LEADER_MODE=$lm DPMM_BENCH_BACKEND=gloo DPMM_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 2975$lm bench.py --gpus 8 --points ${1:-1e6} --steps 40 --warmup 5 2>&1 | tail -1 > /tmp/out8_$lm.json
  python -c "import json; d=json.load(open('/tmp/out8_$lm.json')); print('leader_mode=$lm', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms/step', {k: round(v,3) for k,v in d['host_ms_per_step'].items()})"
done
