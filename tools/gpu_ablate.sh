# conv1b (fused) timing experiments: bench.py --single-mode under D2FE_ABLATE values; prints value, ms/step, conv1b ms per launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-abl}; mkdir -p $O
shift
for ab in "$@"; do
  echo "== D2FE_ABLATE=$ab" >> $O/abl.txt
  D2FE_ABLATE=$ab timeout 300 python bench.py --single-mode --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline_netvlad']['ms_per_call'])" >> $O/abl.txt 2>&1
done
cat $O/abl.txt
