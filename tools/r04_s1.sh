#!/bin/bash
# round-4 session 1 (GPU box): LDS record ring (VPT_LDS_RING) -- bit-identity, A/B against the default library, section cycles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$PWD/volumetric-path-tracer_amd
echo "== bit-identity suites on the ring variant"
VPT_LIB_PATH=$L/libvpt_hip_ring.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_ref.py tests/test_gpu_edge.py tests/test_gpu_scenes.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -4
echo "== full-size suite (default library; 2-iteration batches with the caches asserted)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q 2>&1 | grep -v "amdgpu.ids" | tail -6
echo "== A/B c2"
STEPS=10 bash tools/variants_bench.sh c2 64 default ring default ring
for r in 4 16; do echo "VPT_REGEN_MIN=$r"; VPT_REGEN_MIN=$r STEPS=10 bash tools/variants_bench.sh c2 64 ring; done
for t in 40 56; do echo "VPT_TRANS_MIN=$t"; VPT_TRANS_MIN=$t STEPS=10 bash tools/variants_bench.sh c2 64 ring; done
echo "== A/B c3, c5"
STEPS=3 bash tools/variants_bench.sh c3 64 default ring
STEPS=2 bash tools/variants_bench.sh c5 32 default ring
echo "== c5 with / without the ground-table variants"
STEPS=2 bash tools/variants_bench.sh c5 32 default
VPT_NO_DIR_TABLE=1 STEPS=2 bash tools/variants_bench.sh c5 32 default
echo "== per-frame"
for v in default ring; do
  if [ "$v" = default ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$L/libvpt_hip_$v.so; fi
  python bench.py --config c2 --no-cpu-baseline --no-other-configs --steps 5 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', json.dumps(d.get('per_frame')))"
done
unset VPT_LIB_PATH
echo "== section cycles"
for v in prof ringprof; do
  echo "-- $v"; VPT_LIB_PATH=$L/libvpt_hip_$v.so timeout 600 python tools/perf_probe2.py --config c2 --spp 16 --count-spp 16 2>&1 | grep -v amdgpu.ids | tail -7
done
