#!/bin/bash
# round 6: thresholds re-swept at the final kernels (the walk work is a third smaller than when they were tuned)
cd $GRAFT_REPO_ROOT
TAG=base STEPS=10 bash tools/variants_bench.sh c2 64 default
for t in 36 40 44 52 56 60; do TAG=trans$t STEPS=10 VPT_TRANS_MIN=$t bash tools/variants_bench.sh c2 64 default; done
for t in 2 4 12 16 24 32; do TAG=regen$t STEPS=10 VPT_REGEN_MIN=$t bash tools/variants_bench.sh c2 64 default; done
TAG=base STEPS=10 bash tools/variants_bench.sh c2 64 default
for b in 3 5; do TAG=blocks$b STEPS=10 VPT_BLOCKS_PER_CU=$b bash tools/variants_bench.sh c2 64 default; done
