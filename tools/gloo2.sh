#!/bin/bash
# the N > 1 path on ONE GPU: two ranks share GPU 0 over gloo (RCCL does not put two ranks on one device).  usage: tools/gloo2.sh <tag> [bench args...]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
TAG=$1; shift
O=gpurun_out/run; mkdir -p $O
# --lanes 2 (also bench.py's own choice for --gpus N > 1, LANES_WITH_EXCHANGE; stated because it matters twice here): the two ranks SHARE the device, so each keeps two submits
# in flight (four on the device, as one rank alone would); with four per rank the two processes measured 44 instead of 26 ms per step, with or without the exchange
D2FE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --lanes 2 --single-mode --no-cpu-baseline "$@" > $O/bench_gpus2_$TAG.json 2> $O/bench_gpus2_$TAG.err
grep -v "amdgpu\|socket" $O/bench_gpus2_$TAG.err | tail -3
python - $O/bench_gpus2_$TAG.json <<'PY'
import json, sys
import os
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])          # the line is the LAST line of stdout (bench.py's contract)
f = (j.get("extras") or {}).get("file")
if f and os.path.exists(f): j = json.load(open(f))                           # the full record
print("gpus 2 (gloo, one GPU): value", j["value"], "ms/step", j["ms_per_step"])
for k in ("exchange", "cross_agent", "netvlad_gate"):
    if j.get(k): print(" ", k, json.dumps(j[k])[:1200])
PY
