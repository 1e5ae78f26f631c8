cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
timeout 300 python bench.py --latency-only > $O/lat_graph_pinned.json 2> $O/lat.err
D2FE_GRAPH=0 timeout 300 python bench.py --latency-only > $O/lat_nograph_pinned.json 2>> $O/lat.err
D2FE_GRAPH=0 D2FE_PINNED=0 timeout 300 python bench.py --latency-only > $O/lat_nograph_nopinned.json 2>> $O/lat.err
tail -3 $O/lat.err
python - <<PY
import json
for f in ("lat_graph_pinned","lat_nograph_pinned","lat_nograph_nopinned"):
    try:
        j=json.load(open("$O/%s.json"%f))["latency"]
        print(f, {k.replace("d2fe_","")[:34]: v["p50_ms"] for k,v in j.items() if isinstance(v,dict)})
    except Exception as e: print(f, "failed", e)
PY
