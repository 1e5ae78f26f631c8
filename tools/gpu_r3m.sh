cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "extract_all or graphs" 2>&1 | tail -6
timeout 300 python bench.py --latency-only --latency-calls 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read())['latency']; [print(k, v['p50_ms'], v['p99_ms']) for k,v in j.items() if isinstance(v,dict)]"
