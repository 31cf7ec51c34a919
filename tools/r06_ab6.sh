#!/bin/bash
# round 6: the compiler's instruction scheduler -- -mllvm -amdgpu-sched-strategy=max-ilp (ilp) / max-memory-clause (memcl), -mllvm -amdgpu-use-amdgpu-trackers=1 (trk); whole library rebuilt with the flag
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 default ilp memcl trk; done
STEPS=2 bash tools/variants_bench.sh c5 128 default ilp memcl trk
STEPS=3 bash tools/variants_bench.sh c3 256 default ilp memcl trk
STEPS=2 bash tools/variants_bench.sh c4 128 default ilp memcl trk
