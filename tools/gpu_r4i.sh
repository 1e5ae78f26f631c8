#!/bin/bash
# round 4, batch i: swarm readiness tests (C++ RCCL sequence, one-frame-per-step timeline under gloo) + the default bench
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
timeout 1200 python -m pytest tests/test_cpp_swarm.py tests/test_swarm_gpu.py -x -q -m gpu 2>&1 | tail -12 > $O/pytest_swarm.txt; tail -6 $O/pytest_swarm.txt
bash tools/gpu_r4e.sh
