#!/bin/bash
# round-3 baseline at the round's first commit: section cycles + schedule counters of the tracers (prof variant), GPU tests
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out
export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_prof.so
for c in c2 c3 c4 c5; do
  timeout 300 python tools/perf_probe2.py --config $c --spp 16 > gpurun_out/r03_base_sections_$c.txt 2>&1
done
unset VPT_LIB_PATH
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_base_pytest.txt 2>&1
tail -3 gpurun_out/r03_base_pytest.txt
cat gpurun_out/r03_base_sections_*.txt
