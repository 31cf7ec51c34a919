#!/bin/bash
# round 6: the refill / transition thresholds swept again at the final kernel commit (queue-ordered records + pieces made a refill cheaper); one box
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
echo "# commit ${COMMIT:-unknown}; tools/r06_ab20.sh"
for rep in 1 2; do
TAG=base bash tools/variants_bench.sh c2 64 default
for r in 2 4 6 12 16; do TAG=regen$r VPT_REGEN_MIN=$r bash tools/variants_bench.sh c2 64 default; done
for t in 40 44 52; do TAG=trans$t VPT_TRANS_MIN=$t bash tools/variants_bench.sh c2 64 default; done
TAG="regen4 trans44" VPT_REGEN_MIN=4 VPT_TRANS_MIN=44 bash tools/variants_bench.sh c2 64 default
done
