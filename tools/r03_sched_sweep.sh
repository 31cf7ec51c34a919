#!/bin/bash
# one box: the tracer's scheduling thresholds (VPT_TRANS_MIN / VPT_REGEN_MIN) around their defaults, config 2 bench at spec
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r03_sched_sweep.txt
echo "# commit $(cat .commit_stamp 2>/dev/null); python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline --no-parity --no-per-frame" > $out
run() { v=$(env "$@" timeout 60 python bench.py --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline --no-parity --no-per-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['trace_ms_per_step'])"); echo "$* -> $v" | tee -a $out; }
run A=0
run VPT_TRANS_MIN=40
run VPT_TRANS_MIN=56
run VPT_TRANS_MIN=62
run VPT_REGEN_MIN=4
run VPT_REGEN_MIN=16
run A=1
