#!/bin/bash
# round 4, batch b: the one-launch matcher: parity tests that touch it + timing + phase stamps (development build)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "match or knn or cross or quad or swarm or chain or sliding" 2>&1 | tail -15 > $O/pytest_match.txt
tail -4 $O/pytest_match.txt
timeout 120 python tools/bench_match.py > $O/bench_match.json 2> $O/bench_match.err; cat $O/bench_match.json; tail -3 $O/bench_match.err | grep -v amdgpu.ids
timeout 120 python tools/match_stamps.py 1 64 2>&1 | grep -v amdgpu.ids > $O/match_stamps.txt; cat $O/match_stamps.txt
