# pixel-pair block kernel (netvlad_pair.hip): parity tests, timing against D2FE_NV_PAIR=0, phase stamps, per-kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/nv3; mkdir -p $O; : > $O/r.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k netvlad 2>&1 | tail -15 >> $O/r.txt
for p in 0 1; do
  echo "== D2FE_NV_PAIR=$p" >> $O/r.txt
  D2FE_NV_PAIR=$p timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
for s in ${STEPS:-8 2 14 1}; do timeout 120 python tools/nv_stamps.py $s 32 >> $O/r.txt 2>&1; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
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
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*.db" -delete
cat $R/$O/r.txt
