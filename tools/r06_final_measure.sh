#!/bin/bash
# round-6 measurement session (GPU box).  COMMIT=<id> passed in by the caller.
#   1 full GPU test suite   2 schedule counters + section cycles (c2, c3, c5 direct tracer; c4 schedule + retry statistics)
#   3 rocprofv3 passes of the four configs   4 batch curve / per-frame call   5 the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_pytest_final.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06_pytest_final.txt | tail -5
fi
if [ -z "$SKIP_SCHED" ]; then
for c in c2 c3 c5; do
  (echo "# commit ${COMMIT:-unknown}; python tools/perf_probe2.py --config $c --spp 16 --count-spp 16 (VPT_LIB_PATH: the -DVPT_PROFILE_SECTIONS build for the section cycles)"; VPT_LIB_PATH=$PWD/volumetric-path-tracer_amd/libvpt_hip_prof.so timeout 900 python tools/perf_probe2.py --config $c --spp 16 --count-spp 16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_lanes_sections_${c}.txt
  tail -4 gpurun_out/r06_lanes_sections_${c}.txt
done
(echo "# commit ${COMMIT:-unknown}; python tools/perf_probe2.py --config c4 --spp 16 --count-spp 16 --grid-scale 1.0"; timeout 900 python tools/perf_probe2.py --config c4 --spp 16 --count-spp 16 --grid-scale 1.0 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_lanes_sections_c4.txt
tail -3 gpurun_out/r06_lanes_sections_c4.txt
fi
[ -z "$SKIP_PROFILE" ] && bash tools/r06_profile_all.sh > gpurun_out/r06_profile_all.log 2>&1
if [ -z "$SKIP_CURVE" ]; then
(echo "# commit ${COMMIT:-unknown}; tools/small_launch_probe.sh default (SPPS 1 2 4 8 16 32 64): the per-frame call and the batch curve on one box"; SPPS="1 2 4 8 16 32 64" bash tools/small_launch_probe.sh default 2>&1 | grep -v amdgpu.ids) > gpurun_out/r06_batch_curve.txt
(echo "# the per-frame call with one launch per frame (VPT_NO_FRAME_AHEAD=1, as rounds 1-4)"; VPT_NO_FRAME_AHEAD=1 SPPS=" " bash tools/small_launch_probe.sh default 2>&1 | grep -v amdgpu.ids | head -1
 echo "# raygen over 16-row tiles below 17 iterations instead of below 8 (VPT_RAYGEN_SMALL_ITERS=17)"; VPT_RAYGEN_SMALL_ITERS=17 SPPS="8 16" bash tools/small_launch_probe.sh default 2>&1 | grep -v amdgpu.ids | tail -2
 echo "# independent 8-iteration frames back to back, one / two / three contexts (python tools/short_overlap_probe.py)"; timeout 300 python tools/short_overlap_probe.py --spps 8 2>&1 | grep -v amdgpu.ids) >> gpurun_out/r06_batch_curve.txt
cat gpurun_out/r06_batch_curve.txt
fi
(timeout 900 python bench.py --detail-file gpurun_out/r06_bench_detail.json > gpurun_out/r06_bench_default.out) 2> gpurun_out/r06_bench_default.err
tail -1 gpurun_out/r06_bench_default.out > gpurun_out/r06_bench_default.json
echo "headline line: $(wc -c < gpurun_out/r06_bench_default.json) bytes"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_detail.json"))
r = d["roofline"]
print("c2 %.1f Msamples/s, step %.3f ms" % (d["value"], d["ms_per_step"]), "frac", r["frac"], "kernel frac", r["frac_kernel_issued_fetches"], r.get("valu", {}).get("useful_lane_issue"), d.get("per_frame", {}).get("value"))
for o in d.get("other_configs", []):
    print(o["value"], o["ms_per_step"], o["roofline"]["frac"], json.dumps(o.get("parity"))[:300])
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
