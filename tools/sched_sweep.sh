#!/bin/bash
# one box: the tracer's scheduling thresholds (VPT_TRANS_MIN / VPT_REGEN_MIN) around their defaults; tools/sched_sweep.sh c2:64 c3:64 ...
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for cs in "${@:-c2:64}"; do
  c=${cs%%:*}; s=${cs##*:}
  for e in A=0 VPT_TRANS_MIN=32 VPT_TRANS_MIN=40 VPT_TRANS_MIN=56 VPT_REGEN_MIN=4 VPT_REGEN_MIN=16 VPT_REGEN_MIN=24 A=1; do
    echo -n "$e  "; env $e STEPS=${STEPS:-5} bash tools/variants_bench.sh $c $s default
  done
done
