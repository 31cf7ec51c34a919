#!/bin/bash
# round 6: HBM traffic of config 4's tracer with and without the zero-footprint mask (study library libvpt_hip_zmask.so), one box
cd $GRAFT_REPO_ROOT
export VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_zmask.so
timeout 600 python -m pytest tests/test_gpu_edge.py -x -q -k "zero_footprint" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3
timeout 300 python tools/zmask_ab.py 2>&1 | grep -v amdgpu.ids | tail -9
cd /tmp && export TMPDIR=/tmp
for mode in mask nomask; do
  if [ $mode = mask ]; then export VPT_ZERO_MASK=1; else unset VPT_ZERO_MASK; fi
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_z
    rocprofv3 --pmc $set -d /tmp/rp_z -o r -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame --detail-file /dev/null > /tmp/z.log 2>&1
    DB=$(find /tmp/rp_z -name "*.db" | head -1)
    echo "### $mode pmc: $set"
    python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py pmc $DB 2>&1 | grep -A2 "trace_vol_kernel"
  done
done
