#!/bin/bash
# round 4, batch e: the default bench (pipe-timed headline, batch curve, legs) + wall time
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time; tail -5 $O/bench_default.err | grep -v amdgpu.ids
python - <<PY
import json
j=json.load(open("$O/bench_default.json"))
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"], "conv1b ms", j["roofline"]["avg_launch_ms"], "nv", j["roofline_netvlad"]["ms_per_call"])
for k in ("configs1","exact_mode","fast_mode","device_resident"):
    print(k, j.get(k,{}).get("value"))
for p in j["batch_curve"]["points"]: print(p)
print("stage_ms", j.get("stage_ms"))
print("hbm", json.dumps(j.get("hbm_kernels")))
print("parity", j.get("parity"))
print("wino_vs_exact", {k:v for k,v in j["wino_vs_exact_on_bench_frames"].items() if k!="note"})
print("f16_vs_exact", {k:v for k,v in j["f16x2_vs_exact_on_bench_frames"].items() if k!="note"})
print("lat", {k:v["p50_ms"] for k,v in j["latency"].items() if isinstance(v,dict)})
print("quad", j["quadcam"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
