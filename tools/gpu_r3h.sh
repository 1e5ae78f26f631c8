cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_wino.py -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('split-staging', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
done
