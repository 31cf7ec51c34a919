#!/bin/bash
# GPU box: two SQ counter passes of one bench configuration, per-kernel summary: tools/pmc_quick.sh c2 64
CFG=${1:-c2}; SPP=${2:-64}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf /tmp/rp_q
  rocprofv3 --pmc $set -d /tmp/rp_q -o r -- python $REPO/bench.py --config $CFG --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame > /tmp/pmc_quick.log 2>&1
  DB=$(find /tmp/rp_q -name "*.db" | head -1)
  echo "### pmc: $set"
  python $REPO/profiles/summarize_rocprof.py pmc $DB 2>&1
done
