#!/bin/bash
# round 4, batch g: kernel-trace of the pipe at 4 lanes x 1 stereo frame: who overlaps whom
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
REPO=$PWD; cd /tmp
timeout 200 rocprofv3 --output-format csv --kernel-trace -d $O/trace_pipe4 -o t -- python $REPO/tools/pipe_probe.py --sweep 4x1 --seconds 0.15 > $O/trace_pipe4.log 2>&1
grep stereo_fps $O/trace_pipe4.log | head -2
python $REPO/tools/overlap_stats.py $O/trace_pipe4 0.3 > $O/overlap_pipe4.txt 2>&1; head -4 $O/overlap_pipe4.txt
python - <<PY
import csv,glob
f=glob.glob("$O/trace_pipe4/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
n=len(rows); i0=int(n*0.8); t0=int(rows[i0]['Start_Timestamp'])
out=open("$O/timeline_pipe4.txt","w")
for r in rows[i0:i0+260]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.write("%8.1f %8.1f %7.1f q%-2s g%-7s %s\n"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,r['Queue_Id'],r['Grid_Size_X'],r['Kernel_Name'].replace('d2fe::','').replace('void ','')[:60]))
PY
find $O/trace_pipe4 -name "*.db" -delete; find $O/trace_pipe4 -name "*kernel_trace.csv" -size +3M -delete
