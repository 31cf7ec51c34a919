#!/bin/bash
# claim size study (GPU box; needs a library built with tools/variants/r04_windowed_claims.patch applied, which reads VPT_STUDY_CHUNK): one atomic per VPT_STUDY_CHUNK queue entries, worked through in windows of 256
cd $GRAFT_REPO_ROOT
for ch in ${CHUNKS:-256 1024 4096}; do
  echo "== claim $ch"
  for cs in ${CFGS:-c2:64 c3:64 c5:32 c4:16}; do
    c=${cs%%:*}; s=${cs##*:}
    VPT_STUDY_CHUNK=$ch STEPS=${STEPS:-5} bash tools/variants_bench.sh $c $s default
  done
done
for ch in ${SMALL:-128 512}; do
  echo "== small launches, claim $ch"
  VPT_STUDY_CHUNK=$ch SPPS="1 8" bash tools/small_launch_probe.sh default
done
