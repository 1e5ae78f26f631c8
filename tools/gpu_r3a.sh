# round 3, first GPU pass: full GPU suite (no -x: see every failure), smoke, default bench (incl. the new latency leg)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -80 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest_gpu.txt | tail -60; tail -2 $O/smoke.txt; tail -5 $O/bench.err; head -c 3000 $O/bench.json
