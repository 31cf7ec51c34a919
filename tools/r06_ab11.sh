#!/bin/bash
# round 6: vol tracer -- the retry spins go on while at least N lanes are still spinning (retryN)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=2 bash tools/variants_bench.sh c4 128 default retry8 retry16 retry24; done
