#!/bin/bash
# Container side: stamp the commit the library's sources are at into profiles/kernel_commit.txt (the GPU box's snapshot has no .git), then
# run a command on the GPU box with COMMIT exported:   tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
C=$(git log -1 --format=%h -- volumetric-path-tracer_amd/csrc include volumetric-path-tracer_amd/build.py)
[ -n "$(git status --porcelain -- volumetric-path-tracer_amd/csrc include volumetric-path-tracer_amd/build.py)" ] && C="$C+dirty"
echo "$C" > profiles/kernel_commit.txt
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $T -- "export COMMIT=$C; $*"
