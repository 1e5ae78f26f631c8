# round 3: the driver's default bench line + rocprofv3 kernel-trace stats and PMC passes of the same command (profiles/r03_*)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r3e; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -2 $O/bench_default.err
bash tools/profile.sh r03 wino > $O/profile.log 2>&1
tail -5 $O/profile.log
cp gpurun_out/prof_r03_wino/summary.txt $O/summary.txt 2>/dev/null
head -c 1500 $O/bench_default.json
