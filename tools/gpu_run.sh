#!/bin/bash
# ONE parameterised runner for the GPU box (replaces the per-batch gpu_r*.sh scripts of rounds 2-4).  Tasks run in the order given, separated by `--`:
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh suite -- smoke -- bench -- profile r05 wino'
#
#   suite [pytest args...]       pytest -m gpu (extra args are passed on, e.g.  suite -k "pipe or match")            -> gpurun_out/run/pytest_gpu.txt
#   smoke                        __graft_entry__.smoke()
#   bench [bench.py args...]     bench.py with its JSON line saved and the key numbers printed                        -> gpurun_out/run/bench*.json
#   profile <tag> [precision]    tools/profile.sh: rocprofv3 kernel-trace stats + four PMC passes of bench.py        -> gpurun_out/prof_<tag>_<prec>/summary.txt
#   stats <tag> [bench args...]  tools/prof_stats.sh: kernel-trace stats of any bench configuration
#   ab <tag> [bench args...]     same-box A/B: d2slam_amd/lib/libd2fe_hip_A.so (a copy of the previous build) against the current library,
#                                three alternating runs of bench.py --single-mode (value, ms/step, conv1b ms, NetVLAD ms, per-stage times)
#   envab <tag> VAR a b [args]   the same for an environment switch of the development library
#   latency                      bench.py --latency-only (single-call latencies through the C ABI, p50)
#   pipe <sweep> [args...]       tools/pipe_probe.py --sweep <lanes x frames,...>  (e.g.  pipe 1x1,4x1,8x1 --coalesce 4)
#   layers [args...]             tools/bench_wino.py (per-layer timing of the Winograd kernels, development library)
#   match                        tools/bench_match.py (matchKNN launch time for 1..64 pairs)
#   netvlad [n...]               tools/bench_netvlad.py n... --fused-only + the per-dispatch timeline (tools/nv_timeline.sh)
#   timeline [pipe_probe args]   rocprofv3 kernel trace of the running pipe, per-dispatch timeline of one pass (tools/pipe_timeline.py)
#   py <script> [args...]        any tools/*.py
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
O=$PWD/gpurun_out/run; mkdir -p $O
BENCH_N=0

bench_summary() {      # the key numbers of a bench JSON line
python - "$1" <<'PY'
import json, os, sys
txt = open(sys.argv[1]).read().strip().splitlines()
j = json.loads(txt[-1])                                   # bench.py's contract: the JSON line is the LAST line of stdout
print("line bytes", len(txt[-1]), "| stdout lines", len(txt))
f = (j.get("extras") or {}).get("file")
if f and os.path.exists(f):
    full = json.load(open(f)); os.replace(f, sys.argv[1].replace(".json", "_extras.json")); j = full      # the full record, kept beside the line
r = j.get("roofline") or {}
print("value", j.get("value"), "ms/step", j.get("ms_per_step"), "| conv1b ms", r.get("avg_launch_ms"), "frac", r.get("frac"), "| nv ms", (j.get("roofline_netvlad") or {}).get("ms_per_call"))
for k in ("configs1", "exact_mode", "fast_mode", "device_resident", "quadcam"):
    if j.get(k): print(" ", k, j[k].get("value"), (j[k].get("roofline") or {}).get("frac"))
for k in ("step_roofline", "netvlad_width_sensitivity", "exchange", "index_parity_in_run"):
    if j.get(k): print(" ", k, json.dumps(j[k])[:600])
for p in (j.get("batch_curve") or {}).get("points", []): print("  curve", {k: v for k, v in p.items() if k not in ("note", "host_ms_per_submit_call", "ms_per_submit")})
if j.get("stage_ms"): print("  stage_ms", j["stage_ms"])
if j.get("hbm_kernels"): print("  hbm", json.dumps({k: (v or {}).get("ms_per_launch") for k, v in j["hbm_kernels"].items()}))
if j.get("parity"): print("  parity", {k: v for k, v in j["parity"].items() if k != "note"})
for k in ("wino_vs_exact_on_bench_frames", "f16x2_vs_exact_on_bench_frames"):
    if j.get(k): print(" ", k, {a: b for a, b in j[k].items() if a != "note"})
if j.get("latency"): print("  latency p50", {k: v["p50_ms"] for k, v in j["latency"].items() if isinstance(v, dict)})
if j.get("cpu_baseline"): print("  cpu", j["cpu_baseline"].get("value"), j["cpu_baseline"].get("cores"))
PY
}

ab_line='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], (j.get("roofline_netvlad") or {}).get("ms_per_call"))'

run_task() {
  local t=$1; shift
  echo "=== $t $*"
  case $t in
    suite)   local where=tests; case "$1" in tests/*) where=; ;; esac      # `suite tests/test_x.py [args]` runs that file only
             ( time timeout 2400 python -m pytest $where -x -q -m gpu "$@" 2>&1 | tail -25 ) > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt ;;
    bench)   BENCH_N=$((BENCH_N+1)); local f=$O/bench$BENCH_N
             ( time timeout 1200 python bench.py "$@" > $f.json 2> $f.err ) 2> $f.time; tail -3 $f.time | head -1; grep -v amdgpu.ids $f.err | tail -4; bench_summary $f.json ;;
    profile) bash tools/profile.sh "$@" > $O/profile.log 2>&1; tail -2 $O/profile.log ;;
    stats)   bash tools/prof_stats.sh "$@" ;;
    ab)      local tag=$1; shift; local d=gpurun_out/ab_$tag; mkdir -p $d; : > $d/ab.txt
             for r in 1 2 3; do for lib in A cur; do
               if [ $lib = A ]; then export D2FE_LIB=$PWD/d2slam_amd/lib/libd2fe_hip_A.so; else unset D2FE_LIB; fi
               echo -n "$lib: " >> $d/ab.txt
               timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency --breakdown "$@" 2>$d/err_$lib.txt | python -c "$ab_line" >> $d/ab.txt 2>&1
             done; done; unset D2FE_LIB
             for lib in A cur; do grep "per-stage" $d/err_$lib.txt | tail -1 >> $d/ab.txt; done; cat $d/ab.txt ;;
    envab)   local tag=$1 var=$2 a=$3 b=$4; shift 4; local d=gpurun_out/envab_$tag; mkdir -p $d; : > $d/ab.txt
             for r in 1 2 3; do for v in "$a" "$b"; do
               echo -n "$var=$v: " >> $d/ab.txt
               env $var=$v timeout 300 python bench.py --single-mode --no-cpu-baseline --no-latency --breakdown "$@" 2>$d/err_$v.txt | python -c "$ab_line" >> $d/ab.txt 2>&1
             done; done
             for v in "$a" "$b"; do grep "per-stage" $d/err_$v.txt | tail -1 >> $d/ab.txt; done; cat $d/ab.txt ;;
    latency) timeout 300 python bench.py --latency-only "$@" 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1])["latency"]; print({k:v["p50_ms"] for k,v in j.items() if isinstance(v,dict)})' ;;
    pipe)    local sw=$1; shift; timeout 300 python tools/pipe_probe.py --sweep $sw "$@" 2>/dev/null | grep '^{"coalesce_depth' | python -c '
import sys, json
for l in sys.stdin:
    j = json.loads(l); print("   lanes %d x %d frames (coalesce %s depth %s): %7.1f stereo fps   streams %s" % (j["lanes"], j["frames_per_submit"], j.get("coalesce"), j.get("coalesce_depth"), j["stereo_fps"], json.dumps(j.get("stream_classes"))))' ;;
    layers)  timeout 600 python tools/bench_wino.py "$@" 2>&1 | grep -v amdgpu ;;
    match)   timeout 300 python tools/bench_match.py "$@" 2>&1 | grep -v amdgpu ;;
    netvlad) bash tools/nv_timeline.sh gpurun_out/run/nv ${@:-1 32} ;;
    py)      local s=$1; shift; timeout 900 python tools/$s "$@" 2>&1 | grep -v amdgpu.ids | tail -60 ;;
    timeline) # kernel-trace timeline of ONE pass of the pipe: timeline [pipe_probe args, default --sweep 1x1]
             local d=$PWD/gpurun_out/run/ptl; rm -rf $d; ( export TMPDIR=/tmp; cd /tmp; rocprofv3 --output-format csv --kernel-trace -d $d -o t -- python $OLDPWD/tools/pipe_probe.py --seconds 0.05 ${@:---sweep 1x1} > /dev/null 2>&1 )
             python tools/pipe_timeline.py $d | tee $O/pipe_timeline.txt; rm -rf $d ;;
    *)       echo "unknown task $t"; return 2 ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then [ ${#args[@]} -gt 0 ] && run_task "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_task "${args[@]}"
exit 0
