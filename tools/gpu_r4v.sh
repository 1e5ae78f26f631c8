#!/bin/bash
# round 4, batch v: what NetVLAD costs the one-frame-per-submit pipe: fewer / more hidden-channel groups (development library), NetVLAD grouped over lanes with more lanes
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/r4v; mkdir -p $O
run() { echo "== $*"; timeout 120 python tools/pipe_probe.py "$@" 2>/dev/null | grep '^{"nv_group"' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   lanes %d x %d  group %d coalesce %d : %7.1f fps  host %.3f ms' % (j['lanes'], j['frames_per_submit'], j['nv_group'], j['coalesce'], j['stereo_fps'], j['host_submit_ms']))"; }
{
run --sweep 4x1,6x1,8x1
run --sweep 4x1,8x1 --no-netvlad
run --sweep 8x1,12x1 --nv-group 4
run --sweep 6x1 --nv-group 3
run --sweep 8x1 --nv-group 2
D2FE_NV_BLOCKS=1 D2FE_NV_TAIL_BLOCKS=3 run --dev --sweep 4x1,8x1
D2FE_NV_BLOCKS=128 run --dev --sweep 4x1,8x1
D2FE_NV_BLOCKS=2048 D2FE_NV_TAIL_BLOCKS=48 run --dev --sweep 4x1
run --dev --sweep 4x1
} 2>&1 | tee $O/probe.txt
