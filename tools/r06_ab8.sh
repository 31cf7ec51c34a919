#!/bin/bash
# round 6: interleaved claim cursors in the single-volume direct tracer: 1 (claim1 = rounds 1-5), 8 (default), 16, 32
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 claim1 default claim16 claim32; done
for s in 1 2 4 8 16 32; do for v in claim1 default claim16 claim32; do
  if [ "$v" = default ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_$v.so; fi
  STEPS=40 bash tools/variants_bench.sh c2 $s $v
done; done
STEPS=3 bash tools/variants_bench.sh c3 256 claim1 default claim16
STEPS=2 bash tools/variants_bench.sh c5 128 claim1 default
STEPS=2 bash tools/variants_bench.sh c4 128 claim1 default
