#!/bin/bash
# round 4, batch w: dynamic batching in the pipe (coalesce_depth): parity tests, then fps for callers that keep 1 .. 32 single-frame submits in flight
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r4w; mkdir -p $O
( timeout 900 python -m pytest tests/test_pipe.py -x -q -m gpu 2>&1 | tail -5 ) > $O/pytest_pipe.txt 2>&1; tail -3 $O/pytest_pipe.txt
run() { echo "== $*"; timeout 120 python tools/pipe_probe.py "$@" 2>/dev/null | grep '^{"coalesce_depth"' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   lanes %d  coalesce %d depth %d inflight %2d : %7.1f fps  host %.3f ms' % (j['lanes'], j['coalesce'], j['coalesce_depth'], j['inflight'], j['stereo_fps'], j['host_submit_ms']))"; }
{
for infl in 1 2 4 8 16 32; do run --sweep 4x1 --coalesce 4 --coalesce-depth 2 --inflight $infl; done
for infl in 4 16; do run --sweep 4x1 --coalesce 4 --coalesce-depth 1 --inflight $infl; done
for infl in 4 16 32; do run --sweep 4x1 --coalesce 8 --coalesce-depth 2 --inflight $infl; done
for infl in 16 32; do run --sweep 4x1 --coalesce 4 --coalesce-depth 3 --inflight $infl; done
run --sweep 4x1 --coalesce 4
run --sweep 4x1
run --sweep 1x1
} 2>&1 | tee $O/probe.txt
