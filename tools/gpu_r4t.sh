#!/bin/bash
# round 4, batch t: NetVLAD with the in-launch slab fold: tests, timing with and without (development library switch), per-dispatch timelines
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4t; mkdir -p $O
( timeout 900 python -m pytest tests -x -q -m gpu -k "netvlad or pipe or extract_all or quadcam" 2>&1 | tail -8 ) > $O/pytest_nv.txt 2>&1; tail -4 $O/pytest_nv.txt
D2FE_NV_FOLD=0 timeout 120 python tools/bench_netvlad.py 1 2 4 32 --fused-only 2>&1 | grep NetVLAD > $O/bench_nv_nofold.txt
bash tools/nv_timeline.sh gpurun_out/r4t 1 32 > $O/timeline.log 2>&1
echo "--- slab-sum launches (D2FE_NV_FOLD=0)"; cat $O/bench_nv_nofold.txt; echo "--- fold"; cat $O/bench_nv.txt; head -1 $O/timeline1.txt; tail -1 $O/timeline1.txt; tail -1 $O/timeline32.txt
