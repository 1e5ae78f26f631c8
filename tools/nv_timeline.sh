#!/bin/bash
# per-dispatch timeline of one NetVLAD call (rocprofv3 kernel trace of tools/bench_netvlad.py) for the batch sizes given: tools/nv_timeline.sh <out dir> 1 32
cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
timeout 120 python tools/bench_netvlad.py 1 2 4 32 --fused-only 2>&1 | grep NetVLAD > $O/bench_nv.txt
export TMPDIR=/tmp
for n in "$@"; do
  ( cd /tmp; timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof$n -o t -- python $GRAFT_REPO_ROOT/tools/bench_netvlad.py $n --fused-only > $GRAFT_REPO_ROOT/$O/prof_log$n.txt 2>&1 )
  python - $O/prof$n $n <<'PY' > $O/timeline$n.txt
import csv, glob, sys
O, n = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(O + "/**/t_kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if 'nv_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = rows[0]['Kernel_Name']          # a call starts with the first block's kernel (nv_fpair_kernel, or the generic front kernels at other widths)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'] == first and (i == 0 or rows[i - 1]['Kernel_Name'] != first)]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
print("one NetVLAD call of %s image(s): %d launches" % (n, b - a))
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    print(r['Kernel_Name'].replace('void d2fe::', '')[:64].ljust(64), 'grid', r['Grid_Size_X'].rjust(7), r['Grid_Size_Y'].rjust(3), r['Grid_Size_Z'].rjust(3), 'lds', r['LDS_Block_Size'].rjust(6),
          'vgpr', r['VGPR_Count'], r['Accum_VGPR_Count'], 'start %8.1f' % ((int(r['Start_Timestamp']) - t0) / 1e3), 'dur %7.1f us' % d)
print('sum of kernel durations: %.1f us; span to the next call %.1f us' % (tot, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
  rm -rf $O/prof$n
done
cat $O/bench_nv.txt; for n in "$@"; do tail -1 $O/timeline$n.txt; done
