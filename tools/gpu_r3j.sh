cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py tests/test_wino.py tests/test_reference_golden.py -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --latency-only --latency-calls 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read())['latency']; print({k.replace('d2fe_','')[:30]: v['p50_ms'] for k,v in j.items() if isinstance(v,dict)})"
timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step', j['value'], j['ms_per_step'], j['roofline']['frac'], j['stage_ms']['select'], j['stage_ms']['softmax_cand'])"
