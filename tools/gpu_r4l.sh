#!/bin/bash
# round 4, batch l: sensitivity of the one-frame-per-submit rate to the NetVLAD launch sequence (27 small launches per frame)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for nv in "" "--no-netvlad" "--nv-inline"; do
  timeout 100 python tools/pipe_probe.py --seconds 0.8 $nv --sweep 1x1,2x1,4x1,6x1,4x2,1x32,2x32 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$nv','lanes',r['lanes'],'F',r['frames_per_submit'],'fps',r['stereo_fps'])"
done
