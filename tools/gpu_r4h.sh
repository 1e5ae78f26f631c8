#!/bin/bash
# round 4, batch h: index-parity evidence over 1056 images (tools/mode_disagreement.py) + the parity tests of matcher and pipe
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "match or knn or cross or pipe" 2>&1 | tail -3
timeout 900 python tools/mode_disagreement.py > $O/mode_disagreement.json 2> $O/mode_disagreement.err; tail -8 $O/mode_disagreement.err | grep -v amdgpu
