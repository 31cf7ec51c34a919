#!/bin/bash
# pool tracer vs lane-bound tracer: parity tests with the pool as default, then timings of both (GPU box)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out
OUT=gpurun_out/r03_pool_check.txt
: > $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_vs_ref.py tests/test_gpu_edge.py -x -q -m gpu >> $OUT 2>&1
echo "pytest rc=$?" >> $OUT
for c in ${CFGS:-c2 c3 c5}; do
  spp=16; [ $c = c5 ] && spp=8
  for t in lanes pool; do
    for w in ${WAVES:-10}; do
      [ $t = lanes ] && [ $w != ${WAVES%% *} ] && [ -n "$WAVES" ] && continue
      echo "== $c tracer=$t waves=$w" >> $OUT
      VPT_TRACER=$t VPT_POOL_WAVES=$w timeout 300 bash tools/variants_bench.sh $c $spp default >> $OUT 2>&1
    done
  done
done
tail -40 $OUT
