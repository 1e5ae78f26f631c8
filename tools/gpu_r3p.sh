cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err; tail -c 600 gpurun_out/r3p_bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3p_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['roofline']['frac'], j['latency'].get('process'))
for k,v in j['latency'].items():
    if isinstance(v,dict) and 'p50_ms' in v: print(k, v['p50_ms'])
PY
