#!/bin/bash
# round 6, experiment: the tracer's queue as PIECES of consecutive records (vpt_device.h) vs one entry per ray (VPT_PIECE_MAX=0); same library, one box
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
echo "# commit ${COMMIT:-unknown}; tools/r06_ab18.sh $*"
if [ "$1" = sweep ]; then
for rep in 1 2; do
TAG="VPT_PIECE_MAX=0" VPT_PIECE_MAX=0 bash tools/variants_bench.sh c2 64 default
for pm in 256 384 512 768; do for dv in 8 32; do
  TAG="VPT_PIECE_MAX=$pm VPT_PIECE_DIV=$dv" VPT_PIECE_MAX=$pm VPT_PIECE_DIV=$dv bash tools/variants_bench.sh c2 64 default
done; done; done
exit
fi
for rep in 1 2; do
for pm in 0 512 1024 2048; do
  TAG="VPT_PIECE_MAX=$pm" VPT_PIECE_MAX=$pm bash tools/variants_bench.sh c2 64 default
done
done
for pm in 0 1024; do for dv in 4 8 16; do
  TAG="VPT_PIECE_MAX=$pm VPT_PIECE_DIV=$dv" VPT_PIECE_MAX=$pm VPT_PIECE_DIV=$dv bash tools/variants_bench.sh c2 64 default
done; done
for pm in 0 1024; do
  TAG="VPT_PIECE_MAX=$pm" VPT_PIECE_MAX=$pm bash tools/variants_bench.sh c2 8 default
  TAG="VPT_PIECE_MAX=$pm" VPT_PIECE_MAX=$pm bash tools/variants_bench.sh c3 64 default
  TAG="VPT_PIECE_MAX=$pm" VPT_PIECE_MAX=$pm bash tools/variants_bench.sh c2 1 default
done
