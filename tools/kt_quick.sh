#!/bin/bash
# per-kernel time table of one bench configuration (GPU box): tools/kt_quick.sh c4 32 [extra bench args]
CFG=${1:-c2}; SPP=${2:-16}; shift 2
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_kt
rocprofv3 --kernel-trace --stats -d /tmp/rp_kt -o r -- python $REPO/bench.py --config $CFG --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-per-frame "$@" > /tmp/kt.log 2>&1
DB=$(find /tmp/rp_kt -name "*.db" | head -1)
python $REPO/profiles/summarize_rocprof.py kernel $DB 2>&1 | grep -E "vpt::|kernel  "
