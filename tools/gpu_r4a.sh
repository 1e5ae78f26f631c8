#!/bin/bash
# round 4, batch a: kernel-trace of the frames-in-flight probe (K lanes x 1 stereo frame) -> concurrency statistics
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD; cd /tmp
for pt in 1x1 4x1 8x1; do
  timeout 200 rocprofv3 --output-format csv --kernel-trace -d $O/trace_$pt -o t -- python $REPO/tools/fif_probe.py --sweep $pt --steps 80 > $O/trace_$pt.log 2>&1
  python $REPO/tools/overlap_stats.py $O/trace_$pt 0.4 > $O/overlap_$pt.txt 2>&1
  tail -2 $O/trace_$pt.log | head -1
  head -3 $O/overlap_$pt.txt
done
cd $REPO
timeout 100 python tools/fif_probe.py --sweep 1x1,4x1,8x1 --steps 400 > $O/fif2.json 2> $O/fif2.err
cat $O/fif2.json | head -3
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
