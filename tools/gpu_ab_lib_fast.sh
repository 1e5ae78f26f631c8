# same-box A/B of two library builds in another precision mode (default f16x2; f32 = the exact mode): see gpu_ab_lib.sh.  Usage: <tag> [precision]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ablibf}; mkdir -p $O; PREC=${2:-f16x2}
for r in 1 2 3; do
  for lib in A cur; do
    if [ $lib = A ]; then export D2FE_LIB=$PWD/d2slam_amd/lib/libd2fe_hip_A.so; else unset D2FE_LIB; fi
    echo -n "$lib: " >> $O/ab.txt
    timeout 300 python bench.py --precision $PREC --single-mode --no-cpu-baseline 2>$O/err_$lib.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
