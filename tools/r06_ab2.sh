#!/bin/bash
# round 6: the skip loop's lane threshold (study builds skipN: leave the loop once fewer than N lanes stand in an empty node) and what raygen's rejection loop costs (noreject: WRONG streams, timing only)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3
for rep in 1 2; do
STEPS=10 bash tools/variants_bench.sh c2 64 default skip4 skip8 skip16 noreject
done
STEPS=3 bash tools/variants_bench.sh c3 256 default skip4 skip8 skip16
STEPS=2 bash tools/variants_bench.sh c5 128 default skip4 skip8 skip16
STEPS=2 bash tools/variants_bench.sh c4 128 default skip8
echo "== rest of the GPU suite"
timeout 2000 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py --deselect tests/test_gpu_atmosphere.py --durations=12 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -30
