cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wino.py tests/test_ref_pin.py -q -m gpu -k "wino or variant_b" 2>&1 | tail -3
for nt in 0 2; do
D2FE_WINO_NT=$nt timeout 300 python bench.py --latency-only --latency-calls 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read())['latency']; print('WINO_NT=$nt', {k.replace('d2fe_','')[:30]: v['p50_ms'] for k,v in j.items() if isinstance(v,dict)})"
done
timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('auto', j['value'], j['ms_per_step'], j['roofline']['frac'])"
