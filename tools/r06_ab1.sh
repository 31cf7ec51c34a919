#!/bin/bash
# round 6, GPU call: the zero-footprint mask and raygen's footprint, A/B on one box
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_edge.py -x -q -k "zero_footprint or relaid or heads" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -5
echo "== zero mask A/B"
for rep in 1 2; do
TAG=mask   bash tools/variants_bench.sh c4 128 default
TAG=nomask VPT_NO_ZERO_MASK=1 bash tools/variants_bench.sh c4 128 default
done
TAG=mask4  VPT_ZERO_MASK_SHIFT=2 bash tools/variants_bench.sh c4 128 default
TAG=mask16 VPT_ZERO_MASK_SHIFT=4 bash tools/variants_bench.sh c4 128 default
for rep in 1 2; do
TAG=mask   STEPS=3 bash tools/variants_bench.sh c3 256 default
TAG=nomask STEPS=3 VPT_NO_ZERO_MASK=1 bash tools/variants_bench.sh c3 256 default
done
TAG=mask8  STEPS=3 VPT_ZERO_MASK_SHIFT=3 bash tools/variants_bench.sh c3 256 default
TAG=mask-c5   STEPS=2 VPT_ZERO_MASK_MIN_BYTES=0 bash tools/variants_bench.sh c5 128 default
TAG=nomask-c5 STEPS=2 bash tools/variants_bench.sh c5 128 default
TAG=mask-c2   STEPS=10 VPT_ZERO_MASK_MIN_BYTES=0 bash tools/variants_bench.sh c2 64 default
TAG=nomask-c2 STEPS=10 bash tools/variants_bench.sh c2 64 default
echo "== raygen footprint A/B"
for f in rows squares rows squares; do
TAG=$f STEPS=3 VPT_RAYGEN_FOOTPRINT=$f bash tools/variants_bench.sh c3 256 default
TAG=$f STEPS=3 VPT_RAYGEN_FOOTPRINT=$f bash tools/variants_bench.sh c4 128 default
done
TAG=auto STEPS=3 bash tools/variants_bench.sh c3 256 default
TAG=auto STEPS=3 bash tools/variants_bench.sh c4 128 default
