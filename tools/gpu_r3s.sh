# round-3 NetVLAD evidence with the final build: phase stamps of plan steps 8, 2, 14, 0 and the per-dispatch timeline of one 32-image call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O; : > $O/stamps.txt
for s in 8 2 14 0; do timeout 120 python tools/nv_stamps.py $s 32 2>&1 | grep -v "amdgpu.ids" >> $O/stamps.txt; done
timeout 120 python tools/bench_netvlad.py 1 4 32 --fused-only 2>&1 | grep NetVLAD > $O/bench_nv.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_netvlad.py 32 --fused-only > $GRAFT_REPO_ROOT/$O/prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - $O <<'PY' > $O/timeline.txt
import csv, glob, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(O + "/prof/**/t_kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if 'nv_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nv_fpair_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    print(r['Kernel_Name'][:58].ljust(58), 'grid', r['Grid_Size_X'].rjust(7), r['Grid_Size_Y'].rjust(3), r['Grid_Size_Z'].rjust(3), 'lds', r['LDS_Block_Size'].rjust(6),
          'vgpr', r['VGPR_Count'], r['Accum_VGPR_Count'], 'start %8.1f' % ((int(r['Start_Timestamp']) - t0) / 1e3), 'dur %7.1f us' % d)
print('sum of kernel durations: %.1f us; span %.1f us' % (tot, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
rm -rf $O/prof
cat $O/bench_nv.txt $O/timeline.txt; grep "^step" $O/stamps.txt
