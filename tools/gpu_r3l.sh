cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r3l; mkdir -p $O
for ex in fp32 int8 int8-renorm256; do
D2FE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload quadcam --frames 8 --steps 5 --warmup 2 --exchange $ex 2>/dev/null | grep '^{"metric"' > $O/quad_$ex.json
python -c "import json; j=json.load(open('$O/quad_$ex.json')); print('$ex', j['value'], j['cross_agent']['avg_matches_per_view_pair'], j['cross_agent']['block_bytes'], j['netvlad_gate']['rotation_histogram_dir_prev'])"
done
