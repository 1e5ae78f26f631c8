# NetVLAD: first-block tiles per workgroup sweep + parity of the front block; stamps; per-kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/nv7; mkdir -p $O; : > $O/r.txt
D2FE_NV_FRONT_TPW=3 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "layerwise" 2>&1 | tail -3 >> $O/r.txt
for v in "D2FE_NV_FRONT_TPW=1" "D2FE_NV_FRONT_TPW=2" "D2FE_NV_FRONT_TPW=3" "D2FE_NV_FRONT_TPW=4" "D2FE_NV_FRONT_TPW=5"; do
  echo "== $v" >> $O/r.txt
  env $v timeout 120 python tools/bench_netvlad.py 1 32 --fused-only 2>&1 | grep NetVLAD >> $O/r.txt
done
D2FE_NV_FRONT_TPW=4 timeout 120 python tools/nv_stamps.py 0 32 2>&1 | grep -v "amdgpu.ids" >> $O/r.txt
cat $O/r.txt
