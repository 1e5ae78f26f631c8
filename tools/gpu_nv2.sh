# NetVLAD block-kernel experiments: library A (previous build) vs current with D2FE_NV_FLAGS 0..3; phase stamps; per-kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/nv2; mkdir -p $O; : > $O/r.txt
export D2FE_LIB=$PWD/d2slam_amd/lib/libd2fe_hip_A.so
echo "== lib A" >> $O/r.txt; timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
unset D2FE_LIB
for f in 0 1 2 3; do
  echo "== flags $f" >> $O/r.txt
  D2FE_NV_FLAGS=$f timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
for s in 8 2 14; do for f in 0 3; do D2FE_NV_FLAGS=$f timeout 120 python tools/nv_stamps.py $s 32 >> $O/r.txt 2>&1; done; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
for f in 0 3; do
  D2FE_NV_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --stats -d $R/$O/prof$f -o t -- python $R/tools/bench_netvlad.py 32 --fused-only > /dev/null 2>&1
  python - <<PY >> $R/$O/r.txt
import csv, glob
fs = glob.glob("$R/$O/prof$f/**/t_kernel_stats.csv", recursive=True)
print("== kernel stats flags $f")
tot = 0
for r in csv.DictReader(open(fs[0])):
    if "nv_" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14)); tot += float(r["TotalDurationNs"])
print("sum per call: %.1f us" % (tot / 35 / 1e3))
PY
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*.db" -delete
cat $R/$O/r.txt
