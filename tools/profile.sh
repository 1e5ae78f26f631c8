#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.  Usage: tools/profile.sh <tag> <precision>
TAG=${1:-r02}; PREC=${2:-wino}
OUT=$PWD/gpurun_out/prof_${TAG}_${PREC}
mkdir -p $OUT
export TMPDIR=/tmp
# one submit in flight (--lanes 1): every kernel has the device to itself, as in the `roofline` object of the default bench line
BENCH="python $PWD/bench.py --steps 5 --warmup 2 --precision $PREC --single-mode --no-cpu-baseline --lanes 1"
REPO=$PWD; cd /tmp
# the stats pass runs bench.py with its DEFAULT steps/warmup (the command whose JSON line carries roofline.avg_launch_ms)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --precision $PREC --single-mode --no-cpu-baseline --lanes 1 > $OUT/bench_trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- $BENCH > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $BENCH > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_lds -o p -- $BENCH > $OUT/bench_pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -30
ls -la $OUT/*
for f in $OUT/*.log; do echo "--- $f"; tail -3 $f; done
python $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# keep only small artefacts (gpurun_out is capped at 64 MiB)
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT
