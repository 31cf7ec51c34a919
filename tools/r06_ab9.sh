#!/bin/bash
# round 6: queue entries per claim (VPT_CHUNK_ENTRIES) now that a claim is cheap (8 interleaved cursors)
cd $GRAFT_REPO_ROOT
for s in 2 8 16 64; do for c in 0 32 64 128 256; do
  if [ $c = 0 ]; then unset VPT_CHUNK_ENTRIES; else export VPT_CHUNK_ENTRIES=$c; fi
  TAG="chunk$c" STEPS=30 bash tools/variants_bench.sh c2 $s default
done; done
unset VPT_CHUNK_ENTRIES
for c in 0 64 128; do if [ $c = 0 ]; then unset VPT_CHUNK_ENTRIES; else export VPT_CHUNK_ENTRIES=$c; fi; TAG="chunk$c" STEPS=2 bash tools/variants_bench.sh c5 128 default; TAG="chunk$c" STEPS=3 bash tools/variants_bench.sh c3 256 default; done
