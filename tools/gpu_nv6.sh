# NetVLAD: env sweeps (tail groups), stamps, per-kernel trace; no tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/nv6; mkdir -p $O; : > $O/r.txt
for v in "D2FE_NV_TAIL_BLOCKS=768" "D2FE_NV_TAIL_BLOCKS=700" "D2FE_NV_TAIL_BLOCKS=640" "D2FE_NV_TAIL_BLOCKS=560" "D2FE_NV_TAIL_BLOCKS=480"; do
  echo "== $v" >> $O/r.txt
  env $v timeout 120 python tools/bench_netvlad.py 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
for s in ${STEPS:-0 6}; do timeout 120 python tools/nv_stamps.py $s 32 2>&1 | grep -v "ch[1-3] \|we stored\|wd stored\|barrier0\|amdgpu.ids" >> $O/r.txt; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/$O/prof
rocprofv3 --output-format csv --kernel-trace --stats -d $R/$O/prof -o t -- python $R/tools/bench_netvlad.py 32 --fused-only > /dev/null 2>&1
python - <<PY >> $R/$O/r.txt
import csv, glob
fs = glob.glob("$R/$O/prof/**/t_kernel_stats.csv", recursive=True)
print("== kernel stats")
tot = 0
for r in csv.DictReader(open(fs[0])):
    if "nv_" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14)); tot += float(r["TotalDurationNs"])
print("sum per call: %.1f us" % (tot / 35 / 1e3))
PY
rm -rf $R/$O/prof
cat $R/$O/r.txt
