# full GPU suite + smoke (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
cat $O/pytest_gpu.txt; tail -3 $O/smoke.txt
