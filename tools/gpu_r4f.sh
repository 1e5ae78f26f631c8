#!/bin/bash
# round 4, batch f: sparse descriptor head in the lanes at 2-4 images per pass (D2FE_SPARSE_MIN_BATCH=1) vs the dense head; two repetitions, 1 s per point
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
rm -f $O/pipe5.jsonl
for rep in 1 2; do
  for mb in 4 1; do
    D2FE_SPARSE_MIN_BATCH=$mb timeout 100 python tools/pipe_probe.py --seconds 1.0 --sweep 1x1,3x1,4x1,4x2 2>/dev/null | grep -v pipe_probe | sed "s/^{/{\"min_batch\": $mb, /" >> $O/pipe5.jsonl
  done
done
python - <<PY
import json
for l in open("$O/pipe5.jsonl"):
    r=json.loads(l); print("sparse_min_batch", r["min_batch"], "lanes", r["lanes"], "F", r["frames_per_submit"], r["stereo_fps"])
PY
timeout 900 python -m pytest tests -x -q -m gpu -k "match or knn or cross or pipe" 2>&1 | tail -3
timeout 120 python tools/bench_match.py 2>/dev/null
