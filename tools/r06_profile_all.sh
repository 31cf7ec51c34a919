#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes of the bench command for the four BASELINE configs -> gpurun_out/r06_<cfg>_*.txt,
# and profiles/traffic.json rebuilt from them.  COMMIT=<id> must be passed in (stamped into every summary).
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-other-configs --no-per-frame"
tools/profile_bench.sh r06_c2 --config c2 --steps 3 --warmup 1 $X
tools/profile_bench.sh r06_c3 --config c3 --steps 2 --warmup 1 $X
tools/profile_bench.sh r06_c4 --config c4 --steps 1 --warmup 1 $X
tools/profile_bench.sh r06_c5 --config c5 --steps 1 --warmup 1 $X
python tools/make_traffic_json.py gpurun_out/r06_c2_pmc.txt c2 1920 1080 64
python tools/make_traffic_json.py gpurun_out/r06_c3_pmc.txt c3 1920 1080 64
python tools/make_traffic_json.py gpurun_out/r06_c4_pmc.txt c4 1920 1080 64
python tools/make_traffic_json.py gpurun_out/r06_c5_pmc.txt c5 3840 2160 32
cp profiles/traffic.json gpurun_out/r06_traffic.json
ls -la gpurun_out | grep r06_c
