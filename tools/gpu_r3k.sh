# round 3 final: full GPU suite + smoke, the default bench line, rocprofv3 passes of the same command, and the N = 2 JSON lines (two ranks on ONE GPU over gloo:
# functional evidence of the self-launching N > 1 path, not a scaling number)
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r3k; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -14 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.err
bash tools/profile.sh r03 wino > $O/profile.log 2>&1
cp gpurun_out/prof_r03_wino/summary.txt $O/summary.txt 2>/dev/null
D2FE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --single-mode --no-cpu-baseline --frames 8 --steps 5 --warmup 2 > $O/bench_gpus2_gloo.json 2> $O/g2.err
D2FE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload quadcam --frames 8 --steps 5 --warmup 2 --exchange int8 > $O/bench_gpus2_gloo_quadcam_int8.json 2>> $O/g2.err
tail -2 $O/g2.err; head -c 600 $O/bench_gpus2_gloo.json; echo; head -c 300 $O/bench_gpus2_gloo_quadcam_int8.json
