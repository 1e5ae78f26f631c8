# same-box A/B of two builds of the library: d2slam_amd/lib/libd2fe_hip_A.so (a copy of the previous build) against the current one.
# prints value, ms/step, conv1b ms per launch, NetVLAD ms per call for alternating runs of bench.py --single-mode
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ablib}; mkdir -p $O
for r in 1 2 3; do
  for lib in A cur; do
    if [ $lib = A ]; then export D2FE_LIB=$PWD/d2slam_amd/lib/libd2fe_hip_A.so; else unset D2FE_LIB; fi
    echo -n "$lib: " >> $O/ab.txt
    timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency --breakdown 2>$O/err_$lib.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline_netvlad']['ms_per_call'])" >> $O/ab.txt 2>&1
  done
done
grep "per-stage" $O/err_A.txt | tail -1 >> $O/ab.txt; grep "per-stage" $O/err_cur.txt | tail -1 >> $O/ab.txt
cat $O/ab.txt
