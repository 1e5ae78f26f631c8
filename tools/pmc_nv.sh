#!/bin/bash
# usage (GPU box): [PMC_PASSES="2"] tools/pmc_nv.sh <tag>   -- per-dispatch SQ counters of one NetVLAD call (32 images), up to three passes of 8 counters
TAG=${1:-nv}; REPO=$PWD; OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for i in ${PMC_PASSES:-1 2 3}; do
  eval P=\$P$i
  timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $P -d $OUT/p$i -o p -- python $REPO/tools/bench_netvlad.py 32 --fused-only > $OUT/log$i.txt 2>&1
done
cd $REPO
python - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, sys
from collections import defaultdict, OrderedDict
O = sys.argv[1]
per = OrderedDict()
for f in sorted(glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # last complete call: dispatches in order; key by (kernel name, grid) occurrence index from the end
    byd = defaultdict(dict)
    for r in rows:
        if 'nv_' not in r['Kernel_Name']: continue
        byd[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        byd[int(r['Dispatch_Id'])]['_name'] = r['Kernel_Name'][:52] + ' g' + r['Grid_Size']
    ids = sorted(byd)
    fronts = [i for i in ids if 'nv_block_kernel<false, 1' in byd[i]['_name'] or 'nv_fpair_kernel' in byd[i]['_name']]
    a, b = fronts[-2], fronts[-1]
    for k, i in enumerate([i for i in ids if a <= i < b]):
        d = per.setdefault(k, {})
        d.update(byd[i])
for k, d in per.items():
    n = d.pop('_name')
    print(n)
    print('    ' + '  '.join('%s=%.4g' % (c.replace('SQ_', ''), v) for c, v in d.items()))
PY
find $OUT -name "*.db" -delete; find $OUT -size +1M -name "*.csv" -delete
cat $OUT/summary.txt
