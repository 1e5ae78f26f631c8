#!/bin/bash
# round 4, batch q: NetVLAD placement at one stereo frame per submit: side stream / inline / grouped, 2 repetitions, 1 s per point
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${1:-16}
for opt in "" "--nv-inline" "--nv-group 2" "--coalesce 2" "--coalesce 2 --nv-inline" "--coalesce 4 --nv-inline"; do
  for rep in 1 2; do
  timeout 100 python tools/pipe_probe.py --seconds 1.0 $opt --sweep 2x1,3x1,4x1,6x1,8x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
print('hwq $GPU_MAX_HW_QUEUES [$opt]', ' '.join('%dx1: %.0f' % (json.loads(l)['lanes'], json.loads(l)['stereo_fps']) for l in sys.stdin))"
  done
done
