#!/bin/bash
# round 4, batch d: hardware queues x lanes x NetVLAD placement at one stereo frame per submit
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
rm -f $O/pipe4.jsonl
for q in 4 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 100 python tools/pipe_probe.py --seconds 0.3 --sweep 3x1,4x1,5x1,6x1,8x1 2>/dev/null | grep -v pipe_probe >> $O/pipe4.jsonl
  GPU_MAX_HW_QUEUES=$q timeout 100 python tools/pipe_probe.py --seconds 0.3 --nv-inline --sweep 3x1,4x1,6x1,8x1,12x1 2>/dev/null | grep -v pipe_probe >> $O/pipe4.jsonl
done
python - <<PY
import json
for l in open("$O/pipe4.jsonl"):
    r=json.loads(l); print(r["hwq"], "inline" if r["nv_inline"] else "side  ", r["lanes"], r["stereo_fps"])
PY
