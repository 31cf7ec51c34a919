#!/bin/bash
# round 5, GPU call 2: the GPU suite on the pipelined-tail library, A/B against VPT_NO_ASYNC_TAIL=1 on all four configs, one default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_run2; mkdir -p $O
( timeout 1000 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 ) > $O/pytest.txt
bench1() {  # name cfg spp steps [ENV=VAL ...]
  local name=$1 cfg=$2 spp=$3 steps=$4; shift 4
  env "$@" python bench.py --config $cfg --spp $spp --no-cpu-baseline --no-other-configs --no-per-frame --steps $steps --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['per_sample']
print('%-10s %s spp%s: %9.1f Msamples/s  step %8.3f ms  raygen %7.3f trace %8.3f tail %7.3f  | sum of kernels %8.3f' % ('$name', '$cfg', '$spp', d['value'], d['ms_per_step'], r['raygen_ms_per_step'], r['trace_ms_per_step'], r['tail_resolve_ms_per_step'], r['raygen_ms_per_step'] + r['trace_ms_per_step'] + r['tail_resolve_ms_per_step']))"
}
echo "== async tail A/B (one box)" > $O/ab.txt
for rep in 1 2; do
  bench1 onestream c2 64 10 VPT_NO_ASYNC_TAIL=1 >> $O/ab.txt 2>&1
  bench1 pipelined c2 64 10 VPT_X=1 >> $O/ab.txt 2>&1
done
bench1 onestream c3 256 3 VPT_NO_ASYNC_TAIL=1 >> $O/ab.txt 2>&1
bench1 pipelined c3 256 3 VPT_X=1 >> $O/ab.txt 2>&1
bench1 onestream c5 128 2 VPT_NO_ASYNC_TAIL=1 >> $O/ab.txt 2>&1
bench1 pipelined c5 128 2 VPT_X=1 >> $O/ab.txt 2>&1
bench1 onestream c4 128 2 VPT_NO_ASYNC_TAIL=1 >> $O/ab.txt 2>&1
bench1 pipelined c4 128 2 VPT_X=1 >> $O/ab.txt 2>&1
( timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); echo "bench rc $?" >> $O/ab.txt
tail -12 $O/pytest.txt; cat $O/ab.txt; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r05_run2/bench_default.json"))
    print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "cold", d["roofline"].get("cold_view_msamples_per_s"))
    print("per_frame", d.get("per_frame"))
    print("c1", d.get("c1_cpu_single_thread"))
    print("weak/strong", d.get("weak"), d.get("strong"))
    for o in d.get("other_configs", []):
        print(o["config"]["workload"][:40], o["value"], o["ms_per_step"], o["roofline"]["frac"], o["roofline"].get("frac_void"), o["parity"])
    print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="parity_note"})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r05_run2/bench_default.err").read()[-3000:])
PY
