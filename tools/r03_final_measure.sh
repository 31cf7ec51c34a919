#!/bin/bash
# round-3 measurement session (GPU box).  COMMIT=<id> passed in by the caller.
#   1 full GPU test suite   2 schedule counters over a FULL record chunk (c2, c4 at the bench's grid)   3 rocprofv3 passes of the four configs
#   4 the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r03_pytest_final.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r03_pytest_final.txt | tail -5
if [ -z "$SKIP_SCHED" ]; then
for c in c2 c4; do
  gs=1.0
  (echo "# commit ${COMMIT:-unknown}; python tools/perf_probe2.py --config $c --spp 16 --count-spp 64 --grid-scale $gs"; timeout 900 python tools/perf_probe2.py --config $c --spp 16 --count-spp 64 --grid-scale $gs 2>&1 | grep -v amdgpu.ids) > gpurun_out/r03_lanes_sections_${c}_full.txt
  tail -3 gpurun_out/r03_lanes_sections_${c}_full.txt
done
fi
[ -z "$SKIP_PROFILE" ] && bash tools/r03_profile_all.sh > gpurun_out/r03_profile_all.log 2>&1
(timeout 900 python bench.py > gpurun_out/r03_bench_default.json) 2> gpurun_out/r03_bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_default.json"))
print("c2 %.1f Msamples/s, step %.3f ms" % (d["value"], d["ms_per_step"]), d["roofline"].get("valu", {}).get("useful_lane_issue"), d.get("per_frame", {}).get("value"))
for o in d.get("other_configs", []):
    print(o["value"], o["ms_per_step"], json.dumps(o.get("parity"))[:300])
print(json.dumps(d["config"].get("sky_ground_table")))
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
