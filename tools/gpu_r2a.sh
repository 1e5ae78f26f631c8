set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "netvlad or variant_a" 2>&1 | tail -15 > gpurun_out/r2a/pytest_nv.txt
timeout 600 python -m pytest tests/test_ref_pin.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2a/pytest_ref.txt
timeout 300 python tools/bench_netvlad.py 1 4 32 > gpurun_out/r2a/bench_nv.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_netvlad.py 32 --fused-only > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r2a/kernel_stats.txt
import csv, glob
fs = glob.glob("gpurun_out/r2a/prof/**/t_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
for r in rows[:40]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.3f ms total" % (float(r["TotalDurationNs"]) / 1e6)).rjust(18), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14))
PY
find gpurun_out/r2a/prof -name "*kernel_trace.csv" -size +2M -delete; find gpurun_out/r2a/prof -name "*.db" -delete
cat gpurun_out/r2a/pytest_nv.txt gpurun_out/r2a/pytest_ref.txt gpurun_out/r2a/bench_nv.txt gpurun_out/r2a/kernel_stats.txt
