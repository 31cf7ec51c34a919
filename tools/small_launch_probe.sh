#!/bin/bash
# small launches (the per-frame call, short batches) under library variants: tools/small_launch_probe.sh default c64 ...
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for v in "$@"; do
  if [ "$v" = default ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_$v.so; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --frames 128 --detail-file /dev/null 2>/dev/null | grep '^BENCH_DETAIL ' | cut -c14- | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['per_frame']; print('%-8s per frame %.4f ms (%s)  64-it step %.3f ms' % ('$v', p['ms_per_frame'], p['kernels_ms_last_frame'], d['ms_per_step']))"
  for spp in ${SPPS:-1 2 4 8 16}; do
    python bench.py --spp $spp --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs --no-per-frame --detail-file /dev/null 2>/dev/null | grep '^BENCH_DETAIL ' | cut -c14- | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s spp %2d: step %.4f ms (raygen %.3f trace %.3f tail %.3f) -> %.0f Msamples/s' % ('$v', $spp, d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step'], d['value']))"
  done
done
