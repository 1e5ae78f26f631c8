#!/bin/bash
# round 4, batch m: NetVLAD slab-sum launches (D2FE_NV_SLABSUM, development library) vs the one-frame-per-submit rate
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for ss in 3 0 100; do
  for rep in 1 2; do
  D2FE_NV_SLABSUM=$ss timeout 100 python tools/pipe_probe.py --dev --seconds 0.8 --sweep 1x1,2x1,4x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('slabsum $ss','lanes',r['lanes'],'F',r['frames_per_submit'],'fps',r['stereo_fps'])"
  done
done
D2FE_NV_SLABSUM=0 timeout 100 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep -v amdgpu | tail -3
timeout 100 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep -v amdgpu | tail -3
