#!/bin/bash
# round-4 A/B helper (GPU box): tools/r04_ab.sh "<variants>" [configs...]   e.g. tools/r04_ab.sh "default ring" c2:64 c3:64 c5:32
# optional: CHECK="<variants>" runs the bit-identity suites on those variants first; PROF="<variants>" prints section cycles (c2)
cd $GRAFT_REPO_ROOT
L=$PWD/volumetric-path-tracer_amd
V="$1"; shift
for v in $CHECK; do
  echo "== bit-identity suites on $v"
  VPT_LIB_PATH=$L/libvpt_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_ref.py tests/test_gpu_edge.py tests/test_gpu_scenes.py tests/test_gpu_vol.py -q -x 2>&1 | grep -v "amdgpu.ids" | tail -4
done
for cs in "$@"; do
  c=${cs%%:*}; s=${cs##*:}
  STEPS=${STEPS:-6} bash tools/variants_bench.sh $c $s $V
done
for v in $PROF; do
  echo "-- sections $v"; VPT_LIB_PATH=$L/libvpt_hip_$v.so timeout 600 python tools/perf_probe2.py --config ${PROFCFG:-c2} --spp 16 --count-spp 16 2>&1 | grep -v amdgpu.ids | tail -7
done
