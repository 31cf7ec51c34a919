#!/bin/bash
# Quick VALU-issue / lane-utilisation PMC pass of one bench configuration (GPU box):
#   tools/pmc_quick.sh c3 16      -> gpurun_out/pmc_quick_c3.txt
CFG=${1:-c2}; SPP=${2:-16}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_q
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/rp_q -o r -- \
  python $REPO/bench.py --config $CFG --spp $SPP --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame > $REPO/gpurun_out/pmc_quick_$CFG.log 2>&1
DB=$(find /tmp/rp_q -name "*.db" | head -1)
python $REPO/profiles/summarize_rocprof.py pmc $DB > $REPO/gpurun_out/pmc_quick_$CFG.txt 2>&1
python3 - "$REPO/gpurun_out/pmc_quick_$CFG.txt" <<'PY'
import re, sys
cur = None; data = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(\S.*?)\s+\(dispatches: (\d+), avg duration ([\d.]+) us\)', line)
    if m: cur = (m.group(1)[:70], float(m.group(3))); continue
    m = re.match(r'^\s+(\S+)\s+(\d+)\s+per-dispatch\s+([\d.]+)', line)
    if m and cur: data.setdefault(cur, {})[m.group(1)] = float(m.group(3))
for (k, dur), v in data.items():
    if 'SQ_INSTS_VALU' in v and dur > 200:
        valu = v['SQ_INSTS_VALU']
        # issue time per wave64 VALU instruction: 2.2-2.5 cycles for the simple fp32 / integer classes, 4.2 for min/max/cvt/shift/
        # mul24/3-operand integer/f64/packed/DPP/SGPR-operand forms, 8.1 for transcendentals (profiles/r02_valu_issue_probe.txt)
        print("%-72s %9.0f us  VALU issue busy >= %3.0f%% (every instruction at 2.3 cycles) .. %3.0f%% (at 4.2)  lanes/instr %4.1f" % (
            k, dur, 100 * valu * 2.3 / (1024 * dur * 1e-6 * 2.4e9), 100 * valu * 4.2 / (1024 * dur * 1e-6 * 2.4e9), v['SQ_THREAD_CYCLES_VALU'] / valu))
PY
