#!/bin/bash
# round 4, batch p: which launches cost the one-frame-per-submit rate: matcher (the only SuperPoint-side kernel with scratch), NetVLAD
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for opt in "--no-netvlad" "--no-netvlad --no-match" "--no-match" ""; do
  for rep in 1 2; do
  timeout 100 python tools/pipe_probe.py --seconds 0.8 $opt --sweep 1x1,2x1,4x1 2>/dev/null | grep -v pipe_probe | python -c "
import sys,json
print('[$opt]', ' '.join('%dx1: %.0f' % (json.loads(l)['lanes'], json.loads(l)['stereo_fps']) for l in sys.stdin))"
  done
done
