# NetVLAD: workgroup-target / slab-sum threshold sweep with the pixel-pair kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/nv8; mkdir -p $O; : > $O/r.txt
for v in "D2FE_NV_BLOCKS=512" "D2FE_NV_BLOCKS=384" "D2FE_NV_BLOCKS=448" "D2FE_NV_BLOCKS=640" "D2FE_NV_BLOCKS=768" "D2FE_NV_SLABSUM=2" "D2FE_NV_SLABSUM=4" "D2FE_NV_SLABSUM=5 D2FE_NV_BLOCKS=640"; do
  echo "== $v" >> $O/r.txt
  env $v timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
cat $O/r.txt
