#!/bin/bash
# PMC comparison of the two direct tracers on one bench configuration (GPU box): tools/pmc_pool.sh c2 16
# the pool tracer lives in a STUDY library: python volumetric-path-tracer_amd/build.py --variant pool --with-pool
export VPT_LIB_PATH=${VPT_LIB_PATH:-$(cd "$(dirname "$0")/.." && pwd)/volumetric-path-tracer_amd/libvpt_hip_pool.so}
CFG=${1:-c2}; SPP=${2:-16}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for t in ${TRACERS:-lanes pool}; do
  for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
    rm -rf /tmp/rp_q
    VPT_TRACER=$t rocprofv3 --pmc $set -d /tmp/rp_q -o r -- \
      python $REPO/bench.py --config $CFG --spp $SPP --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame > /tmp/pmc_pool.log 2>&1
    DB=$(find /tmp/rp_q -name "*.db" | head -1)
    echo "=== $CFG tracer=$t" 
    python $REPO/profiles/summarize_rocprof.py pmc $DB 2>&1 | grep -A8 "trace_"
  done
done > $REPO/gpurun_out/pmc_pool_$CFG.txt 2>&1
cat $REPO/gpurun_out/pmc_pool_$CFG.txt
