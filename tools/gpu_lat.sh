cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/${1:-lat}; mkdir -p $O; REPO=$PWD
export TMPDIR=/tmp; cd /tmp
python $REPO/tools/lat_probe.py 60 > $O/probe_plain.txt 2>&1
rocprofv3 --output-format csv --kernel-trace -d $O/trace -o t -- python $REPO/tools/lat_probe.py 30 > $O/probe_traced.txt 2>&1
python $REPO/tools/lat_timeline.py $O/trace 15 > $O/timeline.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
cat $O/probe_plain.txt; cat $O/timeline.txt
