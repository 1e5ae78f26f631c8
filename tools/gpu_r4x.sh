#!/bin/bash
# round 4, final evidence from one build: [full GPU suite + smoke,] the default bench line, rocprofv3 stats + PMC passes of the bench (tools/profile.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
[ "$1" = "nosuite" ] || bash tools/gpu_full.sh r4_final
bash tools/gpu_r4e.sh > gpurun_out/r4/bench_default.summary.txt 2>&1; head -8 gpurun_out/r4/bench_default.summary.txt
bash tools/profile.sh r04 wino > gpurun_out/r4/profile.log 2>&1; tail -2 gpurun_out/r4/profile.log
