#!/bin/bash
# one box: the reciprocal-multiply quotient of to_unit against the division (bit-identity tests), then the bench with and without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r03_fastdiv.txt
echo "# commit $(cat .commit_stamp 2>/dev/null)" > $out
timeout 45 python -m pytest tests/test_gpu_edge.py -k quotient -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee -a $out
run() { v=$(env "$@" timeout 30 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-per-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], 'trace', d['roofline']['trace_ms_per_step'], 'c3/c4/c5', [round(o['value'], 1) for o in d['other_configs']])"); echo "$* -> $v" | tee -a $out; }
run A=fast
run VPT_NO_FAST_DIV=1
