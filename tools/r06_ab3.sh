#!/bin/bash
# round 6: the skip loop's lane threshold, finer sweep (default = 8 now; skipN: study builds), the loop's round cap, the refill / transition thresholds on top of it, the tail at 5 waves per SIMD
cd $GRAFT_REPO_ROOT
echo "== config 2, threshold sweep (skip1 = rounds 1-5's loop)"
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 skip1 skip6 default skip10 skip12 loop4 loop16 tail5; done
echo "== config 2, thresholds of the product library"
for t in 32 40 56; do TAG=trans$t STEPS=10 VPT_TRANS_MIN=$t bash tools/variants_bench.sh c2 64 default; done
for t in 4 12 16; do TAG=regen$t STEPS=10 VPT_REGEN_MIN=$t bash tools/variants_bench.sh c2 64 default; done
echo "== config 5 / 3 / 4"
STEPS=2 bash tools/variants_bench.sh c5 128 skip1 skip6 default skip10 skip12 loop16
STEPS=3 bash tools/variants_bench.sh c3 256 skip1 skip6 default skip12
STEPS=2 bash tools/variants_bench.sh c4 128 skip1 default skip12
echo "== exactness: the GPU suite with the product library (threshold 8)"
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py --durations=8 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -16
