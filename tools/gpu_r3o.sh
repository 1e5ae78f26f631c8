cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --latency-only --latency-calls 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read())['latency']; [print('lat-only', k, v['p50_ms']) for k,v in j.items() if isinstance(v,dict) and ('all' in k or 'fused' in k)]"
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']); [print('full', k, v['p50_ms']) for k,v in j['latency'].items() if isinstance(v,dict) and ('all' in k or 'fused' in k)]"
