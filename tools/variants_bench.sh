#!/bin/bash
# tools/variants_bench.sh <config> <spp> <variant> [variant...]   ("default" = the product library; others: build.py --variant)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
CFG=$1; SPP=$2; shift 2
for v in "$@"; do
  if [ "$v" = default ]; then unset VPT_LIB_PATH; else export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_$v.so; fi
  python bench.py --config $CFG --spp $SPP --no-cpu-baseline --no-other-configs --no-per-frame --steps ${STEPS:-5} --warmup 1 --detail-file /dev/null 2>/dev/null | grep '^BENCH_DETAIL ' | cut -c14- | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['per_sample']
print('%-10s %s spp%s: %9.1f Msamples/s  step %8.3f ms  raygen %7.3f trace %8.3f tail %7.3f  | steps/sample %.3f  fetches %.3f  %s' % ('$v', '$CFG', '$SPP', d['value'], d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step'], p['tracking_steps'], p['density_fetches'], '${TAG:-}'))"
done
