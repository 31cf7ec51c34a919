#!/bin/bash
# round 6: a delta-tracking walk that has left the box of non-empty leaves and whose line robustly misses the sphere ends the path there (nosx = without, the product library of commit 84ed806)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do STEPS=10 bash tools/variants_bench.sh c2 64 nosx default; done
STEPS=20 bash tools/variants_bench.sh c2 8 nosx default
echo "== exactness"
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6
