cd $GRAFT_REPO_ROOT
O=gpurun_out/nvenv; mkdir -p $O; : > $O/r.txt
for v in "D2FE_NV_SLABSUM=3" "D2FE_NV_SLABSUM=2" "D2FE_NV_SLABSUM=2 D2FE_NV_BLOCKS=768" "D2FE_NV_SLABSUM=2 D2FE_NV_BLOCKS=1024" "D2FE_NV_SLABSUM=3 D2FE_NV_TAIL_BLOCKS=512" "D2FE_NV_SLABSUM=3 D2FE_NV_TAIL_BLOCKS=1024"; do
  echo "== $v" >> $O/r.txt
  env $v timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
cat $O/r.txt
