#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats + PMC passes of the bench command, summarised
# into gpurun_out/<tag>_*.txt (copy the ones to keep into profiles/).
#   tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof flags...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" -d /tmp/rp_$name -o r -- python $REPO/bench.py $ARGS > $OUT/${TAG}_${name}.log 2>&1
  find /tmp/rp_$name -name "*.db" | head -1
}
DB=$(run kt --kernel-trace --stats)
# every summary names the commit it was taken at (COMMIT: passed in by the caller, the GPU box has no .git)
echo "# commit ${COMMIT:-unknown}; command: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" > $OUT/${TAG}_kernel_trace.txt
[ -n "$DB" ] && python $REPO/profiles/summarize_rocprof.py kernel $DB >> $OUT/${TAG}_kernel_trace.txt 2>&1
echo "# commit ${COMMIT:-unknown}; command: rocprofv3 --pmc <set> -- python bench.py $ARGS (one pass per set)" > $OUT/${TAG}_pmc.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-24)
  DB=$(run pmc_$n --pmc $set)
  echo "### pmc: $set" >> $OUT/${TAG}_pmc.txt
  [ -n "$DB" ] && python $REPO/profiles/summarize_rocprof.py pmc $DB >> $OUT/${TAG}_pmc.txt 2>&1
done
tail -1 $OUT/${TAG}_kt.log
