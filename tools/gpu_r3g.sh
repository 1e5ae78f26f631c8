cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
j=json.load(open("$O/bench_default.json"))
print(j["value"], j["ms_per_step"], j["avg_matches_per_pair"], j["roofline"]["frac"], j["roofline"]["traffic"], j["parity"])
print(j["wino_vs_exact_on_bench_frames"])
print(j["matcher_queries_past_first_4_candidates_per_step"], j["matcher_exact_scan_rows_per_step"], j["stage_ms"]["match"])
print("configs1",j["configs1"]["value"],"exact",j["exact_mode"]["value"],"fast",j["fast_mode"]["value"],"quad",j["quadcam"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
timeout 600 python -m pytest tests/test_swarm_gpu.py -q -m gpu 2>&1 | tail -3
