#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py with the NetVLAD leg; prints the top kernels.
R=$PWD; OUT=$R/gpurun_out/prof_nv; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python $R/bench.py --single-mode --no-cpu-baseline --netvlad --steps 10 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
fs = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
for r in rows[:30]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(5), ("%.3f ms total" % (float(r["TotalDurationNs"]) / 1e6)).rjust(18), ("%.1f us avg" % (float(r["AverageNs"]) / 1e3)).rjust(14))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
