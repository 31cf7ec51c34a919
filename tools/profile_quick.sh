#!/bin/bash
# quick PMC of the tracer kernels on the GPU box:
#   tools/profile_quick.sh <tag> "<counters>" <bench args...>
TAG=$1; CNT=$2; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rq_$TAG
rocprofv3 --pmc $CNT -d /tmp/rq_$TAG -o r -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $REPO/gpurun_out/${TAG}.log 2>&1
DB=$(find /tmp/rq_$TAG -name "*.db" | head -1)
python $REPO/profiles/summarize_rocprof.py pmc $DB | grep -A10 "trace_kernel<\|trace_vol_kernel<\|tail_resolve\|raygen_kernel<false" | grep -v "^--"
