#!/bin/bash
# round 6: a delta-tracking walk whose position has left the grid's DOMAIN for good (outside by a margin on an axis it does not move back along) and whose line clears the sphere ends the path (dx); default = ea1222c
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do STEPS=10 bash tools/variants_bench.sh c2 64 default dx; done
STEPS=20 bash tools/variants_bench.sh c2 8 default dx
STEPS=3 bash tools/variants_bench.sh c3 256 default dx
echo "== exactness of dx (VPT_LIB_PATH)"
VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_dx.so timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8
