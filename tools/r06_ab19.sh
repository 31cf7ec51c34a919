#!/bin/bash
# round 6, experiment: the tracer's claims through the scalar memory path (s_atomic_add, s_load_dwordx2) vs the vector one (study library: build.py --variant vclaim -DVPT_VECTOR_CLAIMS); one box
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
echo "# commit ${COMMIT:-unknown}; tools/r06_ab19.sh"
for rep in 1 2 3; do bash tools/variants_bench.sh c2 64 default vclaim; done
for c in c3 c5; do bash tools/variants_bench.sh $c 64 default vclaim; done
bash tools/variants_bench.sh c4 16 default vclaim
for s in 8 2; do bash tools/variants_bench.sh c2 $s default vclaim; bash tools/variants_bench.sh c2 $s default vclaim; done
