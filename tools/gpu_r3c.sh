cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -60 > $O/pytest_gpu.txt
tail -40 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
