cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
for f in "" "--no-overlap"; do
  echo "== bench $f" >> $O/ab.txt
  timeout 300 python bench.py --single-mode --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline_netvlad']['ms_per_call'], j['stage_ms'] if 'stage_ms' in j else '')" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
