#!/bin/bash
# round 5, GPU call 3: gate diagnostics, the tests that changed, frame-ahead + drain threshold + raygen tile threshold on short launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_run3; mkdir -p $O
( timeout 300 python tools/gate_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/gate.txt
( timeout 1200 python -m pytest tests/test_gpu_atmosphere.py tests/test_gpu_edge.py "tests/test_gpu_fullsize.py::test_config5_100_instances_4k_dof_sun_and_sky" tests/test_gpu_bench_ranks.py::test_two_ranks_on_one_gpu -q 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/pytest.txt
L=$PWD/volumetric-path-tracer_amd
pf() {  # name [ENV=VAL ...]: the per-frame call and short batches
  local name=$1; shift
  env "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --frames 256 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['per_frame']; print('%-14s per frame %.4f ms = %.0f Msamples/s (%s)  64-it step %.3f ms' % ('$name', p['ms_per_frame'], p['value'], p['kernels_ms_last_frame'], d['ms_per_step']))"
}
sb() {  # name spp [ENV=VAL ...]
  local name=$1 spp=$2; shift 2
  env "$@" python bench.py --spp $spp --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs --no-per-frame 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s spp %2d: step %.4f ms (raygen %.3f trace %.3f tail %.3f) -> %.0f Msamples/s' % ('$name', $spp, d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step'], d['value']))"
}
{
echo "== the per-frame call (vpt_render + vpt_sync per frame), 256 frames"
pf frame-by-frame VPT_NO_FRAME_AHEAD=1
pf ahead16 VPT_X=1
pf ahead32 VPT_FRAME_AHEAD_MAX=32
pf ahead8 VPT_FRAME_AHEAD_MAX=8
pf ahead16+drain1 VPT_LIB_PATH=$L/libvpt_hip_drain1.so
pf nofa+drain1 VPT_NO_FRAME_AHEAD=1 VPT_LIB_PATH=$L/libvpt_hip_drain1.so
pf nofa+drain8 VPT_NO_FRAME_AHEAD=1 VPT_LIB_PATH=$L/libvpt_hip_drain8.so
pf nofa+drain16 VPT_NO_FRAME_AHEAD=1 VPT_LIB_PATH=$L/libvpt_hip_drain16.so
echo "== short batches"
for spp in 4 8 16 64; do
  sb default $spp VPT_X=1
  sb drain1 $spp VPT_LIB_PATH=$L/libvpt_hip_drain1.so
  sb drain8 $spp VPT_LIB_PATH=$L/libvpt_hip_drain8.so
  sb drain16 $spp VPT_LIB_PATH=$L/libvpt_hip_drain16.so
done
sb rows16<=16 8 VPT_RAYGEN_SMALL_ITERS=17
sb rows16<=16 16 VPT_RAYGEN_SMALL_ITERS=17
sb rows16<=32 32 VPT_RAYGEN_SMALL_ITERS=33
sb default 32 VPT_X=1
sb rows16-all 64 VPT_RAYGEN_SMALL_ITERS=65
echo "== other configs, drain thresholds"
for v in default drain1 drain8; do
  if [ $v = default ]; then E=VPT_X=1; else E=VPT_LIB_PATH=$L/libvpt_hip_$v.so; fi
  env $E python bench.py --config c3 --spp 64 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s c3 spp64: step %.4f ms (raygen %.3f trace %.3f tail %.3f)' % ('$v', d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step']))"
  env $E python bench.py --config c4 --spp 8 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s c4 spp8: step %.4f ms (raygen %.3f trace %.3f tail %.3f)' % ('$v', d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step']))"
done
} > $O/short.txt 2>&1
cat $O/gate.txt; tail -30 $O/pytest.txt; cat $O/short.txt
