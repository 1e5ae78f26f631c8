# round-3 final: default bench, rocprofv3 passes of it, NetVLAD stamps/timeline/PMC -- everything profiles/r03_* quotes, from one build
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r3u_bench.json 2> gpurun_out/r3u_bench.err
bash tools/profile.sh r03 wino > gpurun_out/r3u_profile.log 2>&1
bash tools/gpu_r3s.sh > gpurun_out/r3u_r3s.log 2>&1
PMC_PASSES="1 2" bash tools/pmc_nv.sh r3u > gpurun_out/r3u_pmc.log 2>&1
tail -c 300 gpurun_out/r3u_bench.json; tail -3 gpurun_out/r3s/timeline.txt; head -4 gpurun_out/pmc_r3u/summary.txt | cut -c1-300
