#!/bin/bash
# round 4, batch j: pipe with coalescing: parity tests + throughput at one stereo frame per submit
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_pipe.py -x -q -m gpu 2>&1 | tail -6
export GPU_MAX_HW_QUEUES=16
for c in 1 2 3 4; do
  timeout 100 python tools/pipe_probe.py --seconds 1.0 --coalesce $c --sweep 2x1,3x1,4x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('coalesce',r['coalesce'],'lanes',r['lanes'],'fps',r['stereo_fps'],'ms/submit',r['ms_per_submit'])"
done
