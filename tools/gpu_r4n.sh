#!/bin/bash
# round 4, batch n: batch-invariant NetVLAD: full suite, NetVLAD timing, coalesced pipe
cd $GRAFT_REPO_ROOT
bash tools/gpu_full.sh r4_full4
timeout 100 python tools/bench_netvlad.py 1 2 4 32 --fused-only 2>&1 | grep -v amdgpu | tail -5
export GPU_MAX_HW_QUEUES=16
for c in 1 2 4; do
  timeout 100 python tools/pipe_probe.py --seconds 1.0 --coalesce $c --sweep 2x1,3x1,4x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('coalesce',r['coalesce'],'lanes',r['lanes'],'fps',r['stereo_fps'],'ms/submit',r['ms_per_submit'])"
done
timeout 100 python tools/pipe_probe.py --seconds 1.0 --sweep 4x2,4x4,2x32 2>/dev/null | grep -v pipe_probe
