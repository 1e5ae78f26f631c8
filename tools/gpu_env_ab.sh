# same-box A/B of an environment switch: usage <tag> <VAR> <value A> <value B>; alternates bench.py --single-mode runs, prints value, ms/step,
# conv1b ms per launch, NetVLAD ms per call and the per-stage times of the last run of each
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-envab}; mkdir -p $O; VAR=$2
for r in 1 2 3; do
  for v in "$3" "$4"; do
    echo -n "$VAR=$v: " >> $O/ab.txt
    env $VAR=$v timeout 300 python bench.py --single-mode --no-cpu-baseline --breakdown 2>$O/err_$v.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline_netvlad']['ms_per_call'])" >> $O/ab.txt 2>&1
  done
done
for v in "$3" "$4"; do grep "per-stage" $O/err_$v.txt | tail -1 >> $O/ab.txt; done
cat $O/ab.txt
