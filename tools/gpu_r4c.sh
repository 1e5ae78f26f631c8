#!/bin/bash
# round 4, batch c: the frames-in-flight pipe: parity test + throughput sweep (shared device vs disjoint compute units per lane)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_pipe.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_pipe.txt
tail -6 $O/pytest_pipe.txt
timeout 300 python tools/pipe_probe.py --partition --sweep 2x1,4x1,8x1,16x1,2x2,4x2,8x2,2x4,4x4,8x4,2x8,4x8,2x16,2x32 > $O/pipe2_part.json 2> $O/pipe2_part.err; grep -v pipe_probe $O/pipe2_part.json; tail -3 $O/pipe2_part.err | grep -v amdgpu.ids
