#!/bin/bash
# round 6: N interleaved claim cursors instead of one (claimN)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 default claim4 claim8; done
STEPS=2 bash tools/variants_bench.sh c5 128 default claim4 claim8
STEPS=3 bash tools/variants_bench.sh c3 256 default claim4 claim8
STEPS=2 bash tools/variants_bench.sh c4 128 default claim8
for s in 1 8; do for v in default claim8; do
  if [ "$v" = default ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_$v.so; fi
  TAG="spp$s" STEPS=40 bash tools/variants_bench.sh c2 $s $v
done; done
