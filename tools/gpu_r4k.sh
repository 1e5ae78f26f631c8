#!/bin/bash
# round 4, batch k: lanes whose persistent kernels are sized for a share of the device (no CU mask)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for cus in 0 192 128 96 64; do
  timeout 100 python tools/pipe_probe.py --seconds 0.8 --lane-cus $cus --sweep 2x1,3x1,4x1,6x1,4x2 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('lane_cus',r['lane_cus'],'lanes',r['lanes'],'F',r['frames_per_submit'],'fps',r['stereo_fps'])"
done
