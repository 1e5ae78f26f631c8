# round-2 batch B: new-feature GPU tests + bench + NetVLAD group sweep
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_swarm_gpu.py tests/test_quadcam_chain.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest_new.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for b in 256 1024; do D2FE_NV_BLOCKS=$b timeout 120 python tools/bench_netvlad.py 32 --fused-only >> $O/nv_groups.txt 2>&1; done
cat $O/pytest_new.txt; tail -5 $O/bench.err; cat $O/bench.json | head -c 6000; cat $O/nv_groups.txt
