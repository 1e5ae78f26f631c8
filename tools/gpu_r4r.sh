#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_full.sh r4_full5
timeout 300 python bench.py --latency-only 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency']
print({k:v['p50_ms'] for k,v in j.items() if isinstance(v,dict)})"
