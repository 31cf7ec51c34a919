#!/bin/bash
# round 6: raygen -- the rejection loop block-wise (default) vs draw by draw (rgloop), the push loop with a lane threshold (rgpushN), the open-lens instantiation at six waves per SIMD (lens6)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do STEPS=10 bash tools/variants_bench.sh c2 64 rgloop default rgpush8 rgpush16; done
STEPS=3 bash tools/variants_bench.sh c3 256 rgloop default rgpush8
STEPS=2 bash tools/variants_bench.sh c4 128 rgloop default
for rep in 1 2; do STEPS=2 bash tools/variants_bench.sh c5 128 rgloop default lens6 rgpush8; done
echo "== exactness of the product library (block-wise raygen)"
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py --deselect tests/test_gpu_atmosphere.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6
echo "== exactness of rgpush8 (VPT_LIB_PATH)"
VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_rgpush8.so timeout 1200 python -m pytest tests/test_gpu_vs_ref.py tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4
