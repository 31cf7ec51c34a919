#!/bin/bash
# round 6: a ratio-tracking walk that has left the box of non-empty leaves is over (single-volume direct tracer, timed instantiations); notrx = every Tr walk pushes on to the root's far side (rounds 1-5)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do STEPS=10 bash tools/variants_bench.sh c2 64 notrx default; done
STEPS=20 bash tools/variants_bench.sh c2 8 notrx default
STEPS=3 bash tools/variants_bench.sh c3 256 notrx default
echo "== exactness"
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6
