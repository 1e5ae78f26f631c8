cd $GRAFT_REPO_ROOT
O=gpurun_out/frames; mkdir -p $O; : > $O/f.txt
for f in 16 32 48 64; do
  echo "== frames $f" >> $O/f.txt
  timeout 300 python bench.py --single-mode --no-cpu-baseline --frames $f 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac_executed'], j['roofline_netvlad']['ms_per_call'])" >> $O/f.txt 2>&1
done
cat $O/f.txt
