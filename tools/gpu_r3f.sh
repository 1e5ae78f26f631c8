cd $GRAFT_REPO_ROOT
for nf in 0 1; do
D2FE_MATCH_NOFALLBACK=$nf timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NOFALLBACK=$nf', j['value'], j['stage_ms']['match'], j['matcher_queries_past_first_4_candidates_per_step'], j['matcher_exact_scan_rows_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py -q -m gpu -k match 2>&1 | tail -4
