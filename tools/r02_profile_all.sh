#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes of the bench command for the four BASELINE configs -> gpurun_out/r02_<cfg>_*.txt
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-other-configs --no-per-frame"
tools/profile_bench.sh r02_c2 --config c2 --steps 3 --warmup 1 $X
tools/profile_bench.sh r02_c3 --config c3 --steps 2 --warmup 1 $X
tools/profile_bench.sh r02_c4 --config c4 --steps 1 --warmup 1 $X
tools/profile_bench.sh r02_c5 --config c5 --steps 1 --warmup 1 $X
ls -la gpurun_out | grep r02_c
