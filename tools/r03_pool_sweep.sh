#!/bin/bash
# pool tracer: schedule counters and a sweep of waves per CU / start threshold (GPU box)
# the pool tracer lives in a STUDY library: python volumetric-path-tracer_amd/build.py --variant pool --with-pool
export VPT_LIB_PATH=${VPT_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/volumetric-path-tracer_amd/libvpt_hip_pool.so}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out
OUT=gpurun_out/r03_pool_sweep.txt
: > $OUT
for c in ${CFGS:-c2 c3}; do
  echo "== $c schedule (pool)" >> $OUT
  VPT_TRACER=pool timeout 300 python tools/perf_probe2.py --config $c --spp 16 2>&1 | grep -v amdgpu.ids >> $OUT
  for w in ${WAVES:-8 10 12}; do
    for m in ${MINL:-16 40 56}; do
      echo "== $c pool waves=$w min_lanes=$m" >> $OUT
      VPT_TRACER=pool VPT_POOL_WAVES=$w VPT_POOL_MIN_LANES=$m timeout 300 bash tools/variants_bench.sh $c 16 default >> $OUT 2>&1
    done
  done
done
cat $OUT
