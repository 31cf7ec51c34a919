#!/bin/bash
# round 6: raygen recognises rays that cross empty nodes only by their LINE missing the (grown) bounding box of the non-empty leaves (default) instead of walking their pushes out (VPT_NO_LEAFBOX=1)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do TAG=pushes VPT_NO_LEAFBOX=1 STEPS=10 bash tools/variants_bench.sh c2 64 default; TAG=linetest STEPS=10 bash tools/variants_bench.sh c2 64 default; done
for rep in 1 2; do TAG=pushes VPT_NO_LEAFBOX=1 STEPS=2 bash tools/variants_bench.sh c5 128 default; TAG=linetest STEPS=2 bash tools/variants_bench.sh c5 128 default; done
echo "== exactness"
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8
