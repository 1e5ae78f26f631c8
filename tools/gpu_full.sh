#!/bin/bash
# what the driver runs at round end: the GPU test suite and the smoke test (usage: tools/gpu_full.sh <tag>)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/${1:-full}; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
