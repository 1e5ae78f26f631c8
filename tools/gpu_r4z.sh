#!/bin/bash
# round 4, batch z: after a change of the post-processing launches: the tests that cover them, single-call latencies, the one-frame pipe
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
( timeout 1200 python -m pytest tests -x -q -m gpu -k "superpoint or sparse or extract or wino or pipe or golden or quadcam or ref_pin or headline or variant or parity" 2>&1 | tail -3 )
timeout 300 python bench.py --latency-only 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['latency']
print({k:v['p50_ms'] for k,v in j.items() if isinstance(v,dict)})"
timeout 120 python tools/pipe_probe.py --sweep 1x1,4x1,8x1 2>/dev/null | grep '^{"coalesce_depth"' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   lanes %d x %d : %7.1f fps' % (j['lanes'], j['frames_per_submit'], j['stereo_fps']))"
