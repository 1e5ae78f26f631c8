#!/bin/bash
# round 4, batch y: the tail kernels (softmax_cand with batched float4 loads, sparse descriptor head with 64 cells per workgroup): suite + per-stage times
cd $GRAFT_REPO_ROOT
bash tools/gpu_full.sh r4_y
timeout 600 python bench.py --no-cpu-baseline --no-latency --no-batch-curve --no-live-traffic 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', j['value'], 'ms', j['ms_per_step']); print('stage_ms', j.get('stage_ms')); print('hbm', json.dumps({k: (v['ms_per_launch'], v['GBps']) for k, v in j['hbm_kernels'].items()}))
print('parity', j['parity']['keypoints_equal'], j['parity']['scores_equal'], j['parity']['desc_max_abs_diff'])"
