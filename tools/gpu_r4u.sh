#!/bin/bash
# round 4, batch u: NetVLAD hidden-channel split policy: per-dispatch timelines with the cost rule on / off and 10 / 16 tail groups
cd $GRAFT_REPO_ROOT
for cfg in "1 30" "0 30" "1 48" "0 48"; do
  set -- $cfg
  export D2FE_NV_GROUP_RULE=$1 D2FE_NV_TAIL_BLOCKS=$2
  bash tools/nv_timeline.sh gpurun_out/r4u/rule$1_tail$2 1 32 > /dev/null 2>&1
  echo "== rule $1 tail blocks $2"; cat gpurun_out/r4u/rule$1_tail$2/bench_nv.txt
done
