#!/bin/bash
# round 6: queue-ordered compact ray records (qrec: the record stands where the sample stands in its raygen block's queue, carries {direction, slot}; the tracer re-generates the Philox block, reads no head)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do STEPS=10 bash tools/variants_bench.sh c2 64 default qrec; done
STEPS=20 bash tools/variants_bench.sh c2 8 default qrec
STEPS=3 bash tools/variants_bench.sh c3 256 default qrec
STEPS=2 bash tools/variants_bench.sh c4 128 default qrec
echo "== exactness of qrec (VPT_LIB_PATH)"
VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_qrec.so timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_ranks.py --deselect tests/test_gpu_atmosphere_vs_ref.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8
