#!/bin/bash
# round 6: transitions also when fewer than N lanes are walking (walkN), get_closest_object of OUTER_SECOND / OUTER_TOP hoisted into one place (hoist)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 default walk8 walk16 walk24 hoist; done
STEPS=2 bash tools/variants_bench.sh c5 128 default walk8 walk16 walk24 hoist
STEPS=3 bash tools/variants_bench.sh c3 256 default walk16 hoist
STEPS=2 bash tools/variants_bench.sh c4 128 default walk8 walk16 walk24
