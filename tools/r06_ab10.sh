#!/bin/bash
# round 6: the next refill's ray records touched ahead of time through global_load_lds (pf)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do STEPS=10 bash tools/variants_bench.sh c2 64 default pf; done
STEPS=20 bash tools/variants_bench.sh c2 8 default pf
STEPS=3 bash tools/variants_bench.sh c3 256 default pf
