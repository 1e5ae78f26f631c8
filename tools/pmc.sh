#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>
TAG=$1; CTRS=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --pmc $CTRS -d $OUT -o p -- python $REPO/bench.py "$@" > $OUT/bench.log 2>&1
python - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$OUT/**/*counter_collection.csv",recursive=True)[0]
agg=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(int)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].replace("void d2fe::","").replace("d2fe::","")[:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in sorted(agg,key=lambda k:-sum(agg[k].values()))[:8]:
    print(k); print("     "+"  ".join("%s=%.4g"%(c,v/cnt[(k,c)]) for c,v in sorted(agg[k].items())))
PY
find $OUT -name "*.db" -delete; find $OUT -size +1M -name "*.csv" -delete
