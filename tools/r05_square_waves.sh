#!/bin/bash
# tools/r05_square_waves.sh -- A/B of raygen's wave footprint (8 x 8 pixel squares vs 64 pixels of a row; EXPERIMENTS.md round 5, entry 16), then the GPU suite on the new library.
#   libvpt_hip_old.so = the sources before the change (build.py --variant old on the parent commit)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
OUT=gpurun_out/r05_square_waves; mkdir -p $OUT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-8s %s: %9.1f Msamples/s  step %8.3f ms  raygen %7.3f trace %8.3f tail %7.3f' % ('$1', '$2', d['value'], d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step']))"; }
run() { # lib config steps
  if [ "$1" = new ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_$1.so; fi
  timeout 120 python bench.py --config $2 --no-cpu-baseline --no-other-configs --no-per-frame --no-c1 --steps $3 --warmup 1 2>/dev/null | tail -1 | line $1 $2
  unset VPT_LIB_PATH
}
{
  run old c2 10; run new c2 10
  run old c5 2; run new c5 2
} > $OUT/ab.txt 2>&1
(timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8) > $OUT/pytest.txt
