# round-3 final evidence: bench.py under rocprofv3 (stats + PMC passes), NetVLAD per-dispatch timeline and PMC (instruction / LDS pass)
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r03 wino > gpurun_out/r3r_profile.log 2>&1
tail -5 gpurun_out/r3r_profile.log
bash tools/gpu_nv.sh r3r > gpurun_out/r3r_nv.log 2>&1
PMC_PASSES="1 2" bash tools/pmc_nv.sh r3r > gpurun_out/r3r_pmc.log 2>&1
tail -30 gpurun_out/nv/timeline.txt
head -12 gpurun_out/pmc_r3r/summary.txt
