#!/bin/bash
# round 4, batch o: netvlad_group: parity + throughput at one stereo frame per submit
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pipe.py -x -q -m gpu 2>&1 | tail -6
export GPU_MAX_HW_QUEUES=16
for g in 1 2 4; do
  timeout 100 python tools/pipe_probe.py --seconds 1.0 --nv-group $g --sweep 4x1,8x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('nv_group',r['nv_group'],'lanes',r['lanes'],'fps',r['stereo_fps'],'ms/submit',r['ms_per_submit'])"
done
timeout 100 python tools/pipe_probe.py --seconds 1.0 --nv-group 3 --sweep 3x1,6x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('nv_group',r['nv_group'],'lanes',r['lanes'],'fps',r['stereo_fps'],'ms/submit',r['ms_per_submit'])"
