cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_wino.py -q -m gpu 2>&1 | tail -5 > $O/pytest_wino.txt; cat $O/pytest_wino.txt
for nt in 1 2; do
  echo "== D2FE_WINO_NT=$nt" >> $O/bench_wino.txt
  D2FE_WINO_NT=$nt timeout 300 python tools/bench_wino.py --imgs 64 --iters 5 >> $O/bench_wino.txt 2>&1
done
cat $O/bench_wino.txt
for nt in 1 2 1 2; do
  D2FE_WINO_NT=$nt timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NT=$nt', j['value'], j['ms_per_step'], j['roofline']['frac'])" | tee -a $O/bench_ab.txt
done
