# NetVLAD iteration on the GPU box: parity tests, fused/legacy timing, per-dispatch timeline of one call
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-nv}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "netvlad" 2>&1 | tail -15 > $O/pytest_nv.txt
timeout 300 python tools/bench_netvlad.py 1 4 32 ${2:-} > $O/bench_nv.txt 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_netvlad.py 32 --fused-only > $GRAFT_REPO_ROOT/$O/prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - $O <<'PY' > $O/timeline.txt
import csv, glob, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(O + "/prof/**/t_kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if 'nv_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one call = from a front kernel (MODE 1) to the next
idx = [i for i, r in enumerate(rows) if 'nv_block_kernel<false, 1' in r['Kernel_Name'] or 'nv_conv0' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    print(r['Kernel_Name'][:58].ljust(58), 'grid', r['Grid_Size_X'].rjust(7), r['Grid_Size_Y'].rjust(3), r['Grid_Size_Z'].rjust(3), 'lds', r['LDS_Block_Size'].rjust(6),
          'vgpr', r['VGPR_Count'], r['Accum_VGPR_Count'], 'start %8.1f' % ((int(r['Start_Timestamp']) - t0) / 1e3), 'dur %7.1f us' % d)
print('sum of kernel durations: %.1f us; span %.1f us' % (tot, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
cat $O/pytest_nv.txt $O/bench_nv.txt $O/timeline.txt
