#!/bin/bash
# GPU box: A/B of the density-grid layouts on one box (tools/variants_bench.sh lines): tools/layout_ab.sh <cfg> <spp> <layout...>
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
CFG=$1; SPP=$2; shift 2
for L in "$@" "$@"; do
  echo "layout $L"
  if [ "$L" = auto ]; then env -u VPT_GRID_LAYOUT -u VPT_RELAID_MIN_BYTES bash tools/variants_bench.sh $CFG $SPP default
  else VPT_RELAID_MIN_BYTES=0 VPT_GRID_LAYOUT=$L bash tools/variants_bench.sh $CFG $SPP default; fi
done
