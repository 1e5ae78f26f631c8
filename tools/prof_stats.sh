#!/bin/bash
# usage: tools/prof_stats.sh <tag> <bench args...>  -- kernel-trace stats only
TAG=$1; shift
OUT=$PWD/gpurun_out/stats_$TAG; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python $REPO/bench.py "$@" > $OUT/bench.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-100s calls=%-5s total_ms=%8.3f avg_us=%9.2f pct=%s"%(r["Name"].replace("d2fe::","")[:100],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,r["Percentage"]))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
